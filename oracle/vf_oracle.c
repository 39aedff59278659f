/*
 * vf_oracle.c -- CPU restatement of vFind's per-read hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing under vfind_amd/ links, loads or calls this file.  It may be used only by tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py, as the checker.
 *
 * What it restates (reference = /root/reference, vfind 0.3.1):
 *   vfo_find_exact            memchr::memmem::find call           src/lib.rs:148
 *   vfo_sg_stats              parasail sg+stats+scan aligner       src/lib.rs:124-135,155-160
 *                             (third-party: parasail-rs 0.7.8 -> libparasail-sys 0.1.10,
 *                              Cargo.toml:16, Cargo.lock:1197-1204,1432-1441; source NOT in
 *                              /root/reference, algorithm restated from its published
 *                              definition)
 *   vfo_threshold             align_threshold_preflight + min_*    src/lib.rs:100-110,260-261
 *   vfo_find_adapter_match    find_adapter_match                   src/lib.rs:141-166
 *   vfo_translate             translate + both tables              src/lib.rs:16-44,52-77,86-95
 *   vfo_spans / vfo_count     the work + reduce closures           src/lib.rs:275-306
 *   vfo_count_reference_shape the same in parallel_fastq's shape   src/lib.rs:271-306 (W workers, one serial reducer)
 *   vfo_count_fastq           File->MultiGzDecoder->Reader chain   src/lib.rs:233-234
 *
 * Pinning: tests/test_oracle_golden.py checks this file against every known-answer vector the
 * reference holds for the path: tests/test_data/toy.fq.gz -> ground_truth.csv (MAGICAL,4) and
 * the four score-vs-threshold unit tests at src/lib.rs:339-369.  Those vectors are tie-free.
 * PARITY UNPINNED for the `length` statistic on reads with co-optimal alignments (ties): no
 * reference test exercises one and parasail cannot be built or imported here.  The tie policy
 * is isolated in the three POLICY_* helpers below.
 */
#define _GNU_SOURCE
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define VFO_NEG (INT_MIN / 4)

typedef struct {
    const uint8_t *prefix;
    int32_t prefix_len;
    const uint8_t *suffix;
    int32_t suffix_len;
    int32_t match, mismatch, gap_open, gap_extend;
    double accept_prefix, accept_suffix;
    int32_t skip_translation;
} vfo_params;

/* ------------------------------------------------------------------ exact search (lib.rs:148) */
/* leftmost byte-exact occurrence, -1 if none; an empty needle matches at 0 (memmem semantics) */
int64_t vfo_find_exact(const uint8_t *seq, int64_t n, const uint8_t *ad, int64_t m) {
    if (m == 0) return 0;
    if (m > n) return -1;
    for (int64_t k = 0; k + m <= n; ++k)
        if (seq[k] == ad[0] && memcmp(seq + k, ad, (size_t)m) == 0) return k;
    return -1;
}

/* ------------------------------------------------------- scoring matrix (lib.rs:236-237) */
/* parasail_matrix_create("ATCG", match, mismatch): identity over A,T,C,G, case-insensitive
 * mapper, every other byte maps to a wildcard row/column scoring 0. */
static inline int vfo_code(uint8_t c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}
static inline int vfo_sub(int a, int b, int match, int mismatch) {
    if (a == 4 || b == 4) return 0;
    return a == b ? match : mismatch;
}

/* ----------------------------------------------------------------- tie policy P (App. B.3) */
/* P1: a gap state takes "open" only when strictly better than "extend". */
static inline int POLICY_gap_opens(int open_score, int ext_score) { return open_score > ext_score; }
/* P2: a cell takes the diagonal when H equals it, else north (F, gap in the read), else west (E). */
enum { FROM_DIAG, FROM_F, FROM_E };
static inline int POLICY_cell_source(int h, int diag, int f) {
    if (h == diag) return FROM_DIAG;
    if (h == f) return FROM_F;
    return FROM_E;
}
/* P3: end cell = best of the last row, ties -> smallest column; a last-column cell replaces it
 * only when strictly greater, ties within the last column -> smallest row. */
static inline int POLICY_end_replaces(int cand, int best) { return cand > best; }

/* ------------------------------------------------ semi-global DP with length (lib.rs:155-160) */
/* query = adapter (rows i=1..m), db = read (columns j=1..n); all four end gaps free; affine gaps
 * open + (k-1)*ext; returns score and the alignment-column count of the chosen optimal path.
 * end_q/end_db are 1-based coordinates of the chosen end cell (for tests).  n==0 or m==0 gives
 * score VFO_NEG (the reference would fail inside parasail; callers treat it as "no match"). */
/* workspace-taking form used by the batch workers (no allocation per read); ws holds 5*(m+1) ints */
static void vfo_sg_stats_ws(const uint8_t *q, int32_t m, const uint8_t *db, int32_t n, int32_t match,
                            int32_t mismatch, int32_t open, int32_t ext, int32_t *score, int32_t *length,
                            int32_t *end_q, int32_t *end_db, int *ws) {
    *score = VFO_NEG; *length = 0; *end_q = 0; *end_db = 0;
    if (m <= 0 || n <= 0) return;
    int *H = ws, *HL = H + (m + 1), *E = HL + (m + 1), *EL = E + (m + 1), *qc = EL + (m + 1);
    for (int i = 0; i < m; ++i) qc[i] = vfo_code(q[i]);
    for (int i = 0; i <= m; ++i) { H[i] = 0; HL[i] = 0; E[i] = VFO_NEG; EL[i] = 0; }
    int best = VFO_NEG, bestL = 0, bq = 0, bd = 0;
    for (int j = 1; j <= n; ++j) {
        int rc = vfo_code(db[j - 1]);
        int hdiag = 0, hldiag = 0;        /* H[0][j-1] */
        int hup = 0, hlup = 0;            /* H[0][j]   */
        int f = VFO_NEG, fl = 0;
        for (int i = 1; i <= m; ++i) {
            int eo = H[i] - open, ee = E[i] - ext, e, el;
            if (POLICY_gap_opens(eo, ee)) { e = eo; el = HL[i] + 1; } else { e = ee; el = EL[i] + 1; }
            int fo = hup - open, fe = f - ext;
            if (POLICY_gap_opens(fo, fe)) { f = fo; fl = hlup + 1; } else { f = fe; fl = fl + 1; }
            int d = hdiag + vfo_sub(qc[i - 1], rc, match, mismatch), dl = hldiag + 1;
            int h = d; if (e > h) h = e; if (f > h) h = f;
            int hl;
            switch (POLICY_cell_source(h, d, f)) {
            case FROM_DIAG: hl = dl; break;
            case FROM_F: hl = fl; break;
            default: hl = el; break;
            }
            hdiag = H[i]; hldiag = HL[i];
            H[i] = h; HL[i] = hl; E[i] = e; EL[i] = el;
            hup = h; hlup = hl;
        }
        if (POLICY_end_replaces(H[m], best)) { best = H[m]; bestL = HL[m]; bq = m; bd = j; }
    }
    for (int i = 1; i < m; ++i)
        if (POLICY_end_replaces(H[i], best)) { best = H[i]; bestL = HL[i]; bq = i; bd = n; }
    *score = best; *length = bestL; *end_q = bq; *end_db = bd;
}

void vfo_sg_stats(const uint8_t *q, int32_t m, const uint8_t *db, int32_t n, int32_t match,
                  int32_t mismatch, int32_t open, int32_t ext, int32_t *score, int32_t *length,
                  int32_t *end_q, int32_t *end_db) {
    int *ws = (int *)malloc(sizeof(int) * 5 * (size_t)((m > 0 ? m : 0) + 1));
    vfo_sg_stats_ws(q, m, db, n, match, mismatch, open, ext, score, length, end_q, end_db, ws);
    free(ws);
}

/* -------------------------------------------------- thresholds (lib.rs:100-110, 260-261) */
/* returns 0 ok, 1 = "Accept alignment threshold must be between 0 and 1." */
int vfo_threshold(double accept, int32_t match, int32_t m, int32_t *skip, int64_t *T) {
    if (accept > 0. && accept < 1.) *skip = 0;
    else if (accept == 1.) *skip = 1;
    else return 1;
    double min_score = accept * (double)match * (double)m;     /* left to right, as :260 */
    *T = (int64_t)floor(min_score) + 1;                          /* score > min  <=>  score >= T */
    return 0;
}

/* ------------------------------------------------- find_adapter_match (lib.rs:141-166) */
/* returns the cut coordinate or -1 (None). */
static int64_t vfo_find_adapter_match_ws(const uint8_t *seq, int64_t n, const uint8_t *ad, int32_t m,
                                         int32_t skip_align, int64_t T, int32_t match, int32_t mismatch,
                                         int32_t open, int32_t ext, int32_t is_prefix, int *ws) {
    int64_t k = vfo_find_exact(seq, n, ad, m);
    if (k >= 0) return is_prefix ? k + m : k;
    if (skip_align) return -1;
    int32_t score, len, eq, ed;
    if (ws) vfo_sg_stats_ws(ad, m, seq, (int32_t)n, match, mismatch, open, ext, &score, &len, &eq, &ed, ws);
    else vfo_sg_stats(ad, m, seq, (int32_t)n, match, mismatch, open, ext, &score, &len, &eq, &ed);
    if (score == VFO_NEG) return -1;
    if ((int64_t)score >= T) {
        if (is_prefix) return len;
        if (len > n) return -1; /* reference underflows a usize here (:160); unreachable in practice */
        return n - len;
    }
    return -1;
}
int64_t vfo_find_adapter_match(const uint8_t *seq, int64_t n, const uint8_t *ad, int32_t m,
                               int32_t skip_align, int64_t T, int32_t match, int32_t mismatch,
                               int32_t open, int32_t ext, int32_t is_prefix) {
    return vfo_find_adapter_match_ws(seq, n, ad, m, skip_align, T, match, mismatch, open, ext, is_prefix, NULL);
}

/* -------------------------------------------------------- translate (lib.rs:16-44,52-95) */
static const char VFO_AA[65] = "KNKNTTTTRSRSIIMI" "QHQHPPPPRRRRLLLL" "EDEDAAAAGGGGVVVV" "*Y*YSSSS*CWCLFLF";
static inline int vfo_tr_index(uint8_t c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
    }
}
/* returns peptide length, or -1 when n % 3 != 0 */
int64_t vfo_translate(const uint8_t *seq, int64_t n, uint8_t *out) {
    if (n % 3 != 0) return -1;
    int64_t k = 0;
    for (int64_t p = 0; p < n; p += 3) {
        int a = vfo_tr_index(seq[p]), b = vfo_tr_index(seq[p + 1]), c = vfo_tr_index(seq[p + 2]);
        out[k++] = (a == 4 || b == 4 || c == 4) ? 'X' : (uint8_t)VFO_AA[a * 16 + b * 4 + c];
    }
    return k;
}

static int vfo_valid_utf8(const uint8_t *s, int64_t n) {
    int64_t i = 0;
    while (i < n) {
        uint8_t c = s[i];
        if (c < 0x80) { i++; continue; }
        int len; uint32_t cp;
        if ((c & 0xE0) == 0xC0) { len = 2; cp = c & 0x1F; }
        else if ((c & 0xF0) == 0xE0) { len = 3; cp = c & 0x0F; }
        else if ((c & 0xF8) == 0xF0) { len = 4; cp = c & 0x07; }
        else return 0;
        if (i + len > n) return 0;
        for (int k = 1; k < len; ++k) { if ((s[i + k] & 0xC0) != 0x80) return 0; cp = (cp << 6) | (s[i + k] & 0x3F); }
        if ((len == 2 && cp < 0x80) || (len == 3 && cp < 0x800) || (len == 4 && (cp < 0x10000 || cp > 0x10FFFF))) return 0;
        if (cp >= 0xD800 && cp <= 0xDFFF) return 0;
        i += len;
    }
    return 1;
}

/* ------------------------------------------------------------------ count table (lib.rs:263) */
typedef struct { uint64_t hash; uint64_t off; uint32_t len; uint64_t count; } vfo_slot;
typedef struct vfo_table {
    vfo_slot *slots; uint64_t cap, n;
    uint8_t *arena; uint64_t arena_cap, arena_top;
} vfo_table;

static uint64_t vfo_hash(const uint8_t *s, int64_t n) {
    uint64_t h = 1469598103934665603ull;
    for (int64_t i = 0; i < n; ++i) { h ^= s[i]; h *= 1099511628211ull; }
    return h | 1;
}
static vfo_table *vfo_table_new(uint64_t cap) {
    vfo_table *t = (vfo_table *)calloc(1, sizeof(*t));
    t->cap = cap; t->slots = (vfo_slot *)calloc(cap, sizeof(vfo_slot));
    t->arena_cap = 1 << 20; t->arena = (uint8_t *)malloc(t->arena_cap);
    return t;
}
static void vfo_table_add(vfo_table *t, const uint8_t *key, int64_t len, uint64_t cnt);
static void vfo_table_grow(vfo_table *t) {
    vfo_slot *old = t->slots; uint64_t oc = t->cap;
    t->cap *= 2; t->slots = (vfo_slot *)calloc(t->cap, sizeof(vfo_slot));
    for (uint64_t i = 0; i < oc; ++i) if (old[i].hash) {
        uint64_t p = old[i].hash & (t->cap - 1);
        while (t->slots[p].hash) p = (p + 1) & (t->cap - 1);
        t->slots[p] = old[i];
    }
    free(old);
}
static void vfo_table_add(vfo_table *t, const uint8_t *key, int64_t len, uint64_t cnt) {
    if ((t->n + 1) * 2 > t->cap) vfo_table_grow(t);
    uint64_t h = vfo_hash(key, len), p = h & (t->cap - 1);
    for (;;) {
        vfo_slot *s = &t->slots[p];
        if (!s->hash) {
            if (t->arena_top + (uint64_t)len > t->arena_cap) {
                while (t->arena_top + (uint64_t)len > t->arena_cap) t->arena_cap *= 2;
                t->arena = (uint8_t *)realloc(t->arena, t->arena_cap);
            }
            memcpy(t->arena + t->arena_top, key, (size_t)len);
            s->hash = h; s->off = t->arena_top; s->len = (uint32_t)len; s->count = cnt;
            t->arena_top += (uint64_t)len; t->n++;
            return;
        }
        if (s->hash == h && s->len == (uint32_t)len && memcmp(t->arena + s->off, key, (size_t)len) == 0) { s->count += cnt; return; }
        p = (p + 1) & (t->cap - 1);
    }
}
void vfo_table_free(vfo_table *t) { if (t) { free(t->slots); free(t->arena); free(t); } }
uint64_t vfo_table_size(const vfo_table *t) { return t->n; }
uint64_t vfo_table_key_bytes(const vfo_table *t) { return t->arena_top; }
/* dump in slot order: keys concatenated, offsets[n+1], counts[n] */
void vfo_table_dump(const vfo_table *t, uint8_t *keys, uint64_t *offsets, uint64_t *counts) {
    uint64_t k = 0, top = 0;
    for (uint64_t i = 0; i < t->cap; ++i) if (t->slots[i].hash) {
        memcpy(keys + top, t->arena + t->slots[i].off, t->slots[i].len);
        offsets[k] = top; counts[k] = t->slots[i].count; top += t->slots[i].len; k++;
    }
    offsets[k] = top;
}

/* ------------------------------------------- per-read function (lib.rs:275-306, App. A) */
typedef struct {
    int32_t skip_p, skip_s; int64_t Tp, Ts;
} vfo_derived;

static int vfo_derive(const vfo_params *p, vfo_derived *d) {
    /* prefix validated first (:239), then suffix (:249) */
    if (vfo_threshold(p->accept_prefix, p->match, p->prefix_len, &d->skip_p, &d->Tp)) return 1;
    if (vfo_threshold(p->accept_suffix, p->match, p->suffix_len, &d->skip_s, &d->Ts)) return 1;
    return 0;
}

static void vfo_read_span(const vfo_params *p, const vfo_derived *d, const uint8_t *seq, int64_t n,
                          int64_t *start, int64_t *end, int *ws) {
    *start = vfo_find_adapter_match_ws(seq, n, p->prefix, p->prefix_len, d->skip_p, d->Tp, p->match,
                                       p->mismatch, p->gap_open, p->gap_extend, 1, ws);
    *end = vfo_find_adapter_match_ws(seq, n, p->suffix, p->suffix_len, d->skip_s, d->Ts, p->match,
                                     p->mismatch, p->gap_open, p->gap_extend, 0, ws);
}

static void vfo_read_count(const vfo_params *p, const vfo_derived *d, const uint8_t *seq, int64_t n,
                           vfo_table *t, uint8_t *scratch, int *ws) {
    int64_t s, e;
    vfo_read_span(p, d, seq, n, &s, &e, ws);
    if (s < 0 || e < 0 || !(s < e)) return;                       /* :288 */
    if (p->skip_translation) {
        if (vfo_valid_utf8(seq + s, e - s)) vfo_table_add(t, seq + s, e - s, 1);   /* :295-296 */
    } else {
        int64_t k = vfo_translate(seq + s, e - s, scratch);       /* :300 */
        if (k >= 0) vfo_table_add(t, scratch, k, 1);
    }
}

typedef struct {
    const vfo_params *p; const vfo_derived *d;
    const uint8_t *bytes; const uint64_t *offs; int64_t lo, hi;
    int64_t *starts, *ends; vfo_table *t;
} vfo_job;

static void *vfo_worker(void *arg) {
    vfo_job *j = (vfo_job *)arg;
    uint8_t *scratch = NULL; int64_t scap = 0;
    int mmax = j->p->prefix_len > j->p->suffix_len ? j->p->prefix_len : j->p->suffix_len;
    int *ws = (int *)malloc(sizeof(int) * 5 * (size_t)(mmax + 1));
    for (int64_t r = j->lo; r < j->hi; ++r) {
        const uint8_t *seq = j->bytes + j->offs[r]; int64_t n = (int64_t)(j->offs[r + 1] - j->offs[r]);
        if (j->starts) vfo_read_span(j->p, j->d, seq, n, &j->starts[r], &j->ends[r], ws);
        if (j->t) {
            if (n / 3 + 1 > scap) { scap = n / 3 + 64; scratch = (uint8_t *)realloc(scratch, (size_t)scap); }
            vfo_read_count(j->p, j->d, seq, n, j->t, scratch, ws);
        }
    }
    free(scratch);
    free(ws);
    return NULL;
}

static int vfo_run(const vfo_params *p, const uint8_t *bytes, const uint64_t *offs, int64_t nreads,
                   int64_t *starts, int64_t *ends, vfo_table *out, int nthreads) {
    vfo_derived d;
    if (vfo_derive(p, &d)) return 1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256]; vfo_job jobs[256];
    for (int k = 0; k < nthreads; ++k) {
        jobs[k].p = p; jobs[k].d = &d; jobs[k].bytes = bytes; jobs[k].offs = offs;
        jobs[k].lo = nreads * k / nthreads; jobs[k].hi = nreads * (k + 1) / nthreads;
        jobs[k].starts = starts; jobs[k].ends = ends;
        jobs[k].t = out ? (k == 0 ? out : vfo_table_new(1 << 12)) : NULL;
    }
    if (nthreads == 1) vfo_worker(&jobs[0]);
    else {
        for (int k = 0; k < nthreads; ++k) pthread_create(&th[k], NULL, vfo_worker, &jobs[k]);
        for (int k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
    }
    if (out) for (int k = 1; k < nthreads; ++k) {      /* serial reduce, as the calling thread does at :292-306 */
        vfo_table *t = jobs[k].t;
        for (uint64_t i = 0; i < t->cap; ++i) if (t->slots[i].hash)
            vfo_table_add(out, t->arena + t->slots[i].off, t->slots[i].len, t->slots[i].count);
        vfo_table_free(t);
    }
    return 0;
}

/* per-read (start,end) as find_adapter_match returns them; -1 = None.  0 ok, 1 = bad threshold */
int vfo_spans(const vfo_params *p, const uint8_t *bytes, const uint64_t *offs, int64_t nreads,
              int64_t *starts, int64_t *ends, int nthreads) {
    return vfo_run(p, bytes, offs, nreads, starts, ends, NULL, nthreads);
}

/* (sequence,count) multiset over a batch of reads.  *out must be freed with vfo_table_free */
int vfo_count(const vfo_params *p, const uint8_t *bytes, const uint64_t *offs, int64_t nreads,
              int nthreads, vfo_table **out) {
    vfo_table *t = vfo_table_new(1 << 12);
    int rc = vfo_run(p, bytes, offs, nreads, NULL, NULL, t, nthreads);
    if (rc) { vfo_table_free(t); *out = NULL; return rc; }
    *out = t; return 0;
}

/* The same table in the reference's own threading shape (lib.rs:271-306, seq_io::parallel_fastq): `nworkers` pool threads
 * run the `work` closure (:275-291: both adapter searches) on record sets, and ONE thread - the caller - runs the reduce
 * closure (:292-306: translate, hash insert) on every record, set by set, in input order.  This is what BASELINE.md 3 calls
 * the CPU baseline's shape; vfo_count above (a table per thread, merged at the end) is kinder to the CPU.  Record sets of
 * kSet reads; a worker takes the next set that has no owner, the reducer waits for the set it needs. */
#include <stdatomic.h>
enum { kSet = 2048 };
typedef struct {
    const vfo_params *p; const vfo_derived *d;
    const uint8_t *bytes; const uint64_t *offs; int64_t nreads, nsets;
    int64_t *starts, *ends;
    atomic_llong next_set;
    atomic_int *done;                                   /* per set */
} vfo_sets;
static void *vfo_set_worker(void *arg) {
    vfo_sets *S = (vfo_sets *)arg;
    int mmax = S->p->prefix_len > S->p->suffix_len ? S->p->prefix_len : S->p->suffix_len;
    int *ws = (int *)malloc(sizeof(int) * 5 * (size_t)(mmax + 1));
    for (;;) {
        const int64_t s = atomic_fetch_add(&S->next_set, 1);
        if (s >= S->nsets) break;
        const int64_t lo = s * kSet, hi = lo + kSet < S->nreads ? lo + kSet : S->nreads;
        for (int64_t r = lo; r < hi; ++r)
            vfo_read_span(S->p, S->d, S->bytes + S->offs[r], (int64_t)(S->offs[r + 1] - S->offs[r]), &S->starts[r], &S->ends[r], ws);
        atomic_store_explicit(&S->done[s], 1, memory_order_release);
    }
    free(ws);
    return NULL;
}
int vfo_count_reference_shape(const vfo_params *p, const uint8_t *bytes, const uint64_t *offs, int64_t nreads,
                              int nworkers, vfo_table **out) {
    vfo_derived d;
    *out = NULL;
    if (vfo_derive(p, &d)) return 1;
    if (nworkers < 1) nworkers = 1;
    if (nworkers > 256) nworkers = 256;
    vfo_sets S;
    S.p = p; S.d = &d; S.bytes = bytes; S.offs = offs; S.nreads = nreads; S.nsets = (nreads + kSet - 1) / kSet;
    S.starts = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nreads + 1));
    S.ends = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nreads + 1));
    S.done = (atomic_int *)calloc((size_t)S.nsets + 1, sizeof(atomic_int));
    atomic_init(&S.next_set, 0);
    pthread_t th[256];
    for (int k = 0; k < nworkers; ++k) pthread_create(&th[k], NULL, vfo_set_worker, &S);
    vfo_table *t = vfo_table_new(1 << 12);
    uint8_t *scratch = NULL; int64_t scap = 0;
    for (int64_t s = 0; s < S.nsets; ++s) {             /* the calling thread: lib.rs:292-306, in input order */
        while (!atomic_load_explicit(&S.done[s], memory_order_acquire)) sched_yield();
        const int64_t lo = s * kSet, hi = lo + kSet < nreads ? lo + kSet : nreads;
        for (int64_t r = lo; r < hi; ++r) {
            const int64_t a = S.starts[r], b = S.ends[r];
            if (a < 0 || b < 0 || !(a < b)) continue;   /* :288 */
            const uint8_t *seq = bytes + offs[r];
            if (p->skip_translation) {
                if (vfo_valid_utf8(seq + a, b - a)) vfo_table_add(t, seq + a, b - a, 1);
            } else {
                const int64_t n = (int64_t)(offs[r + 1] - offs[r]);
                if (n / 3 + 1 > scap) { scap = n / 3 + 64; scratch = (uint8_t *)realloc(scratch, (size_t)scap); }
                const int64_t k = vfo_translate(seq + a, b - a, scratch);
                if (k >= 0) vfo_table_add(t, scratch, k, 1);
            }
        }
    }
    for (int k = 0; k < nworkers; ++k) pthread_join(th[k], NULL);
    free(scratch); free(S.starts); free(S.ends); free(S.done);
    *out = t;
    return 0;
}

/* ---------------------------------------------- FASTQ input chain (lib.rs:233-234, 271) */
/* gzip only (MultiGzDecoder is unconditional at :233; every member is read, CHANGELOG.md:30);
 * strict four-line records.  return: 0 ok, 1 bad threshold, 2 cannot open, 3 not gzip,
 * 4 malformed FASTQ / truncated gzip */
typedef struct { uint8_t *b; size_t n, cap; } vfo_buf;
static int vfo_getline(gzFile f, vfo_buf *l) {
    l->n = 0;
    for (;;) {
        if (l->cap - l->n < 4096) { l->cap = l->cap ? l->cap * 2 : 8192; l->b = (uint8_t *)realloc(l->b, l->cap); }
        if (!gzgets(f, (char *)l->b + l->n, (int)(l->cap - l->n))) return l->n ? 1 : 0;
        l->n += strlen((char *)l->b + l->n);
        if (l->n && l->b[l->n - 1] == '\n') { l->n--; if (l->n && l->b[l->n - 1] == '\r') l->n--; return 1; }
        if (gzeof(f)) return 1;
    }
}
int vfo_count_fastq(const vfo_params *p, const char *path, int nthreads, vfo_table **out, int64_t *nreads_out) {
    *out = NULL;
    vfo_derived d;
    FILE *fp = fopen(path, "rb");
    if (!fp) return 2;
    unsigned char magic[2] = {0, 0};
    size_t got = fread(magic, 1, 2, fp);
    fclose(fp);
    if (vfo_derive(p, &d)) return 1;
    if (got == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) return 3;
    if (got == 1) return 3;
    gzFile f = gzopen(path, "rb");
    if (!f) return 2;
    gzbuffer(f, 1 << 20);
    vfo_buf l = {0, 0, 0};
    uint8_t *bytes = NULL; uint64_t *offs = NULL; size_t bcap = 0, ocap = 0; uint64_t top = 0; int64_t nr = 0;
    int rc = 0;
    for (;;) {
        if (!vfo_getline(f, &l)) break;
        if (l.n == 0) { /* tolerate blank lines at end of input */
            int more = 0; while (vfo_getline(f, &l)) if (l.n) { more = 1; break; }
            if (more) rc = 4;
            break;
        }
        if (l.b[0] != '@') { rc = 4; break; }
        if (!vfo_getline(f, &l)) { rc = 4; break; }
        if (top + l.n + 1 > bcap) { bcap = bcap ? bcap * 2 : (1 << 20); while (top + l.n + 1 > bcap) bcap *= 2; bytes = (uint8_t *)realloc(bytes, bcap); }
        if ((size_t)nr + 2 > ocap) { ocap = ocap ? ocap * 2 : 4096; offs = (uint64_t *)realloc(offs, ocap * sizeof(uint64_t)); }
        memcpy(bytes + top, l.b, l.n); offs[nr] = top; size_t sl = l.n; top += l.n;
        if (!vfo_getline(f, &l) || l.n == 0 || l.b[0] != '+') { rc = 4; break; }
        if (!vfo_getline(f, &l) || l.n != sl) { rc = 4; break; }
        nr++;
    }
    int zerr; gzerror(f, &zerr);
    if (zerr != Z_OK && zerr != Z_STREAM_END) rc = rc ? rc : 4;
    gzclose(f); free(l.b);
    if (!rc) {
        if (!offs) offs = (uint64_t *)malloc(sizeof(uint64_t));
        offs[nr] = top;
        rc = vfo_count(p, bytes ? bytes : (const uint8_t *)"", offs, nr, nthreads, out);
        if (nreads_out) *nreads_out = nr;
    }
    free(bytes); free(offs);
    return rc;
}
