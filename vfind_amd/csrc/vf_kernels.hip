// vf_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the variant-extraction hot path.
//
// One read per lane, 64-lane waves, integer VALU only (no MFMA: there is no dense contraction in
// this path).  Reference semantics being replaced (vfind 0.3.1, src/lib.rs):
//   k_scan     memmem::find for both adapters (lib.rs:148) + the control flow of
//              find_adapter_match (lib.rs:141-166): eight pattern bases at every start position, the
//              whole pattern at the survivors; appends reads that need an alignment to per-workgroup lists
//   k_filter   exact pre-filter of those reads: Myers bit-vector edit distance of the adapter against
//              every column -> the DP windows that can hold an accepted alignment
//   k_dpall    parasail sg_stats_scan semi-global affine DP with the `length` statistic
//              (lib.rs:155-160), tie policy P of SURVEY.md App. B.3, for every task class of the filter in
//              one launch: over a band of ten diagonals (tasks with one unit edit), over the window(s),
//              over the whole read; k_dp = the whole-read DP alone when no filter applies
//   k_finish   seq[start..end] extraction (lib.rs:288-290), translate (lib.rs:16-44) and the
//              HashMap<String,u64> reduce (lib.rs:292-306) as a device hash-count table: one-shot lookup,
//              dense insert list, counts in per-XCD arrays
//   k_generic  the same per-read function byte-wise, for reads/adapters the 2-bit fast path
//              cannot hold (non-ACGT bytes, adapters > 32 bp, reads > 640 bp, exotic scores)
//   k_export / k_merge / k_rehash   table compaction, (key,count) merge for the multi-GPU exchange, growth
#include "vf_device.h"

#include <algorithm>
#include <cstdlib>

namespace vf {

#define WAVE 64
constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbit(hi, lo, sh);   // (hi:lo >> sh)[31:0], sh in 0..31
}

template <typename T> __device__ __forceinline__ T ld_agent(const T *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ void st_agent(T *p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave-aggregated append: returns the slot of this lane in list (valid only if want)
__device__ __forceinline__ uint32_t wave_append(uint32_t *counter, bool want) {
    uint64_t m = __ballot(want);
    uint32_t total = __popcll(m);
    uint32_t base = 0;
    if (total) {
        uint32_t leader = __ffsll((unsigned long long)m) - 1;
        if (lane_id() == leader) base = atomicAdd(counter, total);
        base = __shfl(base, leader);
    }
    uint32_t rank = __popcll(m & ((1ull << lane_id()) - 1ull));
    return base + rank;
}

// wave-aggregated add of popcount(pred) to a workgroup-local (LDS) counter
__device__ __forceinline__ void wave_count_lds(uint32_t *s_counter, bool pred) {
    uint64_t m = __ballot(pred);
    if (m && lane_id() == (uint32_t)(__ffsll((unsigned long long)m) - 1)) atomicAdd(s_counter, (uint32_t)__popcll(m));
}

// ------------------------------------------------------------------------------------------------
// row load: W dwords per read, rows are only 8-byte aligned (W*4 = 40/80/160 bytes)
// ------------------------------------------------------------------------------------------------
// VF_ROWS_NT=1 loads the packed rows non-temporally (tried so that the 420 MB row stream would not push the count table out
// of the 256 MiB Infinity Cache): measured slower everywhere -- cfg 2 1.03 against 0.95 ms per step, k_scan 0.38 against
// 0.33 ms: the rows themselves are re-read by the next kernel of the batch and want the cache
#ifndef VF_ROWS_NT
#define VF_ROWS_NT 0
#endif
template <int W> __device__ __forceinline__ void load_row(const uint32_t *rows, uint64_t r, uint32_t (&x)[W + 2]) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 *p = reinterpret_cast<const u32x2 *>(rows + r * W);
#pragma unroll
    for (int i = 0; i < W / 2; ++i) {
#if VF_ROWS_NT
        const u32x2 v = __builtin_nontemporal_load(p + i);
#else
        const u32x2 v = p[i];
#endif
        x[2 * i] = v.x;
        x[2 * i + 1] = v.y;
    }
    x[W] = 0;
    x[W + 1] = 0;
}

// ------------------------------------------------------------------------------------------------
// Task entries.  A work list of the filter and DP stages holds read indices; bit 31 of an entry says that the read's suffix
// coordinate was still pending when k_scan wrote the span (so that the prefix pass can take a failed prefix's pending suffix
// back, and push an accepted one on, without reading the span: dp_commit only stores).  A batch is at most 2^31 reads.
// (Round 4 also built task RECORDS - the row copied by k_scan next to its index, dense per segment, every stage reading
// records instead of gathering 40-byte rows: 470 MB less through the memory side per cfg-2 step of 2.86 GB, and no faster;
// with 15 % substitutions, where nearly every read is a task and the "gather" is a stream already, 3 % slower.  What the step
// follows is the number of L2 requests, not bytes: EXPERIMENTS R4.4.)
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kTaskSufPending = 1u << 31;
constexpr uint32_t kTaskIndexMask = 0x7FFFFFFFu;
struct TaskSrc { const uint32_t *rows; const uint16_t *lens; int W; };      // the batch the entries point into
struct RecHead { uint32_t r; int n; bool suf_pending; };
__device__ __forceinline__ RecHead task_head(const TaskSrc &src, uint32_t e) {
    const uint32_t r = e & kTaskIndexMask;
    return RecHead{r, (int)src.lens[r], (e & kTaskSufPending) != 0u};
}
__device__ __forceinline__ const uint32_t *task_row(const TaskSrc &src, uint32_t e) { return src.rows + (uint64_t)(e & kTaskIndexMask) * src.W; }

// ------------------------------------------------------------------------------------------------
// exact leftmost search of an m-base pattern in a packed read (memmem::find, lib.rs:148)
//
// Stage 1 compares only (up to) eight bases of each pattern at every start position: XOR-accumulate
// over those bases, acc slot p (2 bits) stays 0 iff read[p+j] == pat[j] for every checked j.  The
// checked bases are pairs (b, b+16) where the pattern is long enough - both sit at the same bit shift
// of the read, one word apart, so a pair costs one funnel shift - and single bases otherwise.
// Stage 2 walks the surviving start positions from the left and compares the whole pattern against
// the read's own bits; the first one that matches is memmem's answer.  A random position survives
// stage 1 with probability 4^-8, so stage 2 almost always runs once per adapter.
// Pattern constants are moved to VGPRs: VALU instructions with an SGPR operand issue at half rate.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t to_vgpr_u(uint32_t s) {
    uint32_t v;
    asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}
struct ScanPlan { int nlo, nhi; };                      // bases b < nlo and b+16 for b < nhi are checked
__device__ __forceinline__ ScanPlan scan_plan(int m) {
    const int pairs = max(0, min(4, m - 16));
    const int lo_room = min(m, 16) - pairs;
    const int singles = max(0, min(8 - 2 * pairs, lo_room));
    return ScanPlan{pairs + singles, pairs};
}
// candidate start positions: one (even) bit per position, 16 positions a word
template <int W> __device__ __forceinline__ void scan_candidates(const uint32_t (&acc)[W], uint32_t (&h)[W]) {
#pragma unroll
    for (int w = 0; w < W; ++w) h[w] = ~(acc[w] | (acc[w] >> 1)) & 0x55555555u;
}
// leftmost candidate, or a value >= 2^30 when there is none (__ffs(0) - 1 = 0xFFFFFFFF)
template <int W> __device__ __forceinline__ uint32_t first_candidate(const uint32_t (&h)[W]) {
    uint32_t pos = 0xFFFFFFFFu;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint32_t pw = (((uint32_t)(__ffs((int)h[w]) - 1)) >> 1) + 16u * (uint32_t)w;
        pos = min(pos, pw);
    }
    return pos;
}
template <int W> __device__ __forceinline__ void drop_candidate(uint32_t (&h)[W], uint32_t pos) {
    const uint32_t bit = 1u << (2u * (pos & 15u));
#pragma unroll
    for (int w = 0; w < W; ++w) h[w] &= ~((pos >> 4) == (uint32_t)w ? bit : 0u);
}
// whole-pattern compare at start position p (p + m <= n <= 16 W).  row = the lane's column of the wave's row stage in LDS
// (word w at row[w * WAVE]): three words at a dynamic offset, which registers cannot give.  Rounds 1-3 re-read them from
// global memory - "L1-hot", except that a CU's 16 KB of L1 holds a fraction of the rows its twenty waves have in flight:
// 3.5 M of k_scan's 7.8 M L2 requests a step and 1.3 M of its 4.6 M memory-side reads were these re-reads (a bare stream of
// the same rows costs 3.1 M requests of 128 bytes and nothing more: scripts/ubench/row_stream.hip, profiles/r04_requests.json)
template <int W> __device__ __forceinline__ bool verify_at(const uint32_t *row, int p, uint64_t pat, int m) {
    const int idx = p >> 4;
    const uint32_t bs = 2u * (uint32_t)(p & 15);
    const uint32_t d0 = row[idx * WAVE];
    const uint32_t d1 = idx + 1 < W ? row[(idx + 1) * WAVE] : 0u;
    const uint32_t d2 = idx + 2 < W ? row[(idx + 2) * WAVE] : 0u;
    const uint32_t lo = funnel(d1, d0, bs), hi = funnel(d2, d1, bs);
    const uint64_t mask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
    const uint64_t diff = (((uint64_t)hi << 32 | lo) ^ pat) & mask;
    return diff == 0ull;
}

template <int W>
__device__ __forceinline__ void exact_find2(const uint32_t (&r)[W + 2], const uint32_t *row, int n, const ScanParams &sp,
                                            int &ps, int &ss) {
    uint32_t accP[W], accS[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { accP[w] = 0; accS[w] = 0; }
    const int mp = (int)sp.pre_len, ms = (int)sp.suf_len;
    const ScanPlan pp = scan_plan(mp), sq = scan_plan(ms);
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const bool p0 = b < pp.nlo, p1 = b < pp.nhi, s0 = b < sq.nlo, s1 = b < sq.nhi;       // wave-uniform
        if (p0 || s0) {
            uint32_t xs[W + 1];
#pragma unroll
            for (int w = 0; w <= W; ++w) xs[w] = b ? funnel(r[w + 1], r[w], 2 * b) : r[w];
            if (p0) {
                const uint32_t bp = to_vgpr_u((uint32_t)((sp.pre_pat >> (2 * b)) & 3ull) * 0x55555555u);
#pragma unroll
                for (int w = 0; w < W; ++w) accP[w] |= xs[w] ^ bp;
            }
            if (p1) {
                const uint32_t bp = to_vgpr_u((uint32_t)((sp.pre_pat >> (2 * b + 32)) & 3ull) * 0x55555555u);
#pragma unroll
                for (int w = 0; w < W; ++w) accP[w] |= xs[w + 1] ^ bp;
            }
            if (s0) {
                const uint32_t bp = to_vgpr_u((uint32_t)((sp.suf_pat >> (2 * b)) & 3ull) * 0x55555555u);
#pragma unroll
                for (int w = 0; w < W; ++w) accS[w] |= xs[w] ^ bp;
            }
            if (s1) {
                const uint32_t bp = to_vgpr_u((uint32_t)((sp.suf_pat >> (2 * b + 32)) & 3ull) * 0x55555555u);
#pragma unroll
                for (int w = 0; w < W; ++w) accS[w] |= xs[w + 1] ^ bp;
            }
        }
    }
    uint32_t hP[W], hS[W];
    scan_candidates<W>(accP, hP);
    scan_candidates<W>(accS, hS);
    ps = -1; ss = -1;
    bool doneP = false, doneS = false;
    for (;;) {
        uint32_t cp = 0, cs = 0;
        if (!doneP) { cp = first_candidate<W>(hP); if (cp + (uint32_t)mp > (uint32_t)n) doneP = true; }     // none, or every later one is further right still
        if (!doneS) { cs = first_candidate<W>(hS); if (cs + (uint32_t)ms > (uint32_t)n) doneS = true; }
        if (!__any(!doneP || !doneS)) break;
        if (!doneP) {
            if (verify_at<W>(row, (int)cp, sp.pre_pat, mp)) { ps = (int)cp; doneP = true; }
            else drop_candidate<W>(hP, cp);
        }
        if (!doneS) {
            if (verify_at<W>(row, (int)cs, sp.suf_pat, ms)) { ss = (int)cs; doneS = true; }
            else drop_candidate<W>(hS, cs);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_scan: exact search for both adapters, span bookkeeping, DP work lists
// ------------------------------------------------------------------------------------------------
// Work lists are segmented per workgroup: block b appends to list[b*seg ..) through a counter in LDS
// and publishes the count at kernel end, so no global counter is ever contended (a single global
// append counter serialises ~10 ns per wave and dominated this kernel).
#ifndef VF_SCAN_WAVES
#define VF_SCAN_ATTR __launch_bounds__(kBlock)
#else
#define VF_SCAN_ATTR __launch_bounds__(kBlock, VF_SCAN_WAVES)      // A/B: register cap (8 -> 64 registers: two waves fit a SIMD beside three k_finish waves)
#endif
template <int W>
__global__ VF_SCAN_ATTR void k_scan(const uint32_t *__restrict__ rows, const uint16_t *__restrict__ lens,
                                                 uint64_t n_reads, ScanParams sp, short2 *__restrict__ spans,
                                                 BatchLists bl) {
    __shared__ uint32_t s_cnt[2];
    // wave-private stage of the trip's rows, one column per lane ([word][lane]): the whole-pattern compare reads three words of
    // it at a dynamic offset (verify_at)
    __shared__ uint32_t s_stage[kBlock / WAVE][WAVE * W];
    uint32_t *stg = s_stage[threadIdx.x / WAVE];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x < 16) bl.tot[threadIdx.x] = 0;      // per-class task totals of the two filter passes
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t n_round = (n_reads + WAVE - 1) / WAVE * WAVE;
    const uint64_t seg0 = (uint64_t)blockIdx.x * bl.seg;
    // the next trip's row is fetched before this trip's search starts
    uint32_t x_next[W + 2];
    int n_next = 0;
    auto prefetch = [&](uint64_t rr) {
        if (rr < n_reads) { load_row<W>(rows, rr, x_next); n_next = lens[rr]; }
        else {
            n_next = 0;
#pragma unroll
            for (int i = 0; i < W + 2; ++i) x_next[i] = 0;
        }
    };
    prefetch((uint64_t)blockIdx.x * kBlock + threadIdx.x);
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n_round; r += stride) {
        const bool valid = r < n_reads;
        uint32_t x[W + 2];
#pragma unroll
        for (int i = 0; i < W + 2; ++i) x[i] = x_next[i];
        const int n = n_next;
        prefetch(r + stride);
        int ps, ss;
        __builtin_amdgcn_wave_barrier();              // (nobody is still comparing against the previous trip's rows: lanes only read their own column, but keep the order explicit)
#pragma unroll
        for (int w = 0; w < W; ++w) stg[w * WAVE + lane_id()] = x[w];
        exact_find2<W>(x, stg + lane_id(), n, sp, ps, ss);
        int start = ps >= 0 ? ps + (int)sp.pre_len : (sp.align_pre ? (int)kPending : (int)kNone);
        int end = ss >= 0 ? ss : ((sp.align_suf && (sp.independent || start != kNone)) ? (int)kPending : (int)kNone);
        const bool needP = valid && start == kPending;
        const bool needS = valid && end == kPending && (sp.independent || start >= 0);
        const uint32_t ip = wave_append(&s_cnt[0], needP);
        if (needP) bl.listP[seg0 + ip] = (uint32_t)r | (end == kPending ? kTaskSufPending : 0u);
        const uint32_t is = wave_append(&s_cnt[1], needS);
        if (needS) bl.listS[seg0 + is] = (uint32_t)r;
        if (valid) spans[r] = make_short2((short)start, (short)end);
    }
    __syncthreads();
    if (threadIdx.x == 0) { bl.cntP[blockIdx.x] = s_cnt[0]; bl.cntS[blockIdx.x] = s_cnt[1]; }
}

// ------------------------------------------------------------------------------------------------
// Semi-global affine-gap DP with the length statistic, one task (read, adapter) per lane.
// Adapter rows live in VGPRs as packed cells  score<<13 | pref<<10 | len  (vf_device.h); the
// substitution term comes from a 4 x M profile in LDS (one ds_read_b128 per 4 rows, bank-conflict
// free with a 36-dword row stride), so the LDS pipe, not the VALU, pays for it.
//
// gfx950 issues add/sub/shift/and/or/xor at full rate but min/max, v_max3, v_bfe, v_mad_u32_u24,
// v_and_or, v_alignbit, v_cndmask, v_cmp at HALF rate (scripts/ubench/valu_rate*.hip), so the cell is
// built from full-rate ops except the three unavoidable maxima:
//   E: add add max and | F: add add max or | diag: add (+LDS) | H: max3 and   = 14 issue units
// Columns are updated in place (a cell is read before it is written), so no cell value is ever copied.
//
// Tie policy P through the 3-bit preference field: candidates enter each max with distinct
// preferences (E: open 0 < extend 1;  F: open 2 < extend 3;  H: diag 4 > F 3 > E 0), the integer max
// therefore picks exactly the policy's winner and carries its length; the field is cleared / set to
// its standing value right after (H, E -> 0; F -> 3).
// ------------------------------------------------------------------------------------------------
constexpr int kProfStride = 36;                      // dwords per base in the LDS profile
constexpr int32_t kClr = (int32_t)~kPrefMask;
constexpr int32_t kV0 = kLenBias;                    // H[0][j] = 0: score 0, pref 0, biased length 0
constexpr int32_t kNegF = (int32_t)(((uint32_t)kNegScore << kScoreShift) | (3u << kPrefShift));
constexpr int32_t kNegE = (int32_t)((uint32_t)kNegScore << kScoreShift);

struct DpLds {
    int32_t prof[4 * kProfStride] __attribute__((aligned(16)));
};
// prof[c][i] = (adapter[i] == c ? (match-mismatch)<<13 : 0) + diag preference 4; call by all threads,
// then __syncthreads()
__device__ __forceinline__ void dp_lds_fill(DpLds &L, const DpParams &dp) {
    for (int idx = threadIdx.x; idx < 4 * kProfStride; idx += kBlock) {
        const int c = idx / kProfStride, i = idx % kProfStride;
        const bool hit = i < 32 && ((dp.eq[c] >> i) & 1u);
        L.prof[idx] = (int32_t)((hit ? dp.delta : 0u) + (4u << kPrefShift));
    }
}

__device__ __forceinline__ int32_t to_vgpr(int32_t s) {
    int32_t v;
    asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}

// one DP column: (Hin,Ein) = column j-1, (Hout,Eout) = column j; pc = profile row of the read base
template <int M>
__device__ __forceinline__ void dp_column(const int32_t (&Hin)[M], const int32_t (&Ein)[M], int32_t (&Hout)[M],
                                          int32_t (&Eout)[M], const int32_t *pc, const DpParams &dp, int32_t &hlast) {
    constexpr int MQ = (M + 3) / 4;
    int32_t pv[4 * MQ];
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
        const int4 v = reinterpret_cast<const int4 *>(pc)[q];
        pv[4 * q] = v.x; pv[4 * q + 1] = v.y; pv[4 * q + 2] = v.z; pv[4 * q + 3] = v.w;
    }
    // the four gap constants sit in VGPRs: VALU instructions with an SGPR operand issue at half rate on
    // gfx950 (scripts/ubench/valu_rate3.hip), and these adds are 4 of the cell's 11 instructions
    const int32_t e_open = to_vgpr(dp.e_open_k), e_ext = to_vgpr(dp.e_ext_k), f_open = to_vgpr(dp.f_open_k), f_ext = to_vgpr(dp.f_ext_k);
    int32_t hdiag = kV0, hup = kV0, f1 = kNegF;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const int32_t hc = Hin[i], ec = Ein[i];
        const int32_t e = max(hc + e_open, ec + e_ext) & kClr;
        const int32_t f = max(hup + f_open, f1 + f_ext) | (int32_t)(1u << kPrefShift);
        const int32_t d = hdiag + pv[i];
        const int32_t h = max(max(d, f), e) & kClr;
        hdiag = hc;
        Hout[i] = h;
        Eout[i] = e;
        hup = h;
        f1 = f;
    }
    hlast = hup;
}

template <int M> __device__ __forceinline__ void dp_init(int32_t (&H)[M], int32_t (&E)[M], const DpParams &dp, bool true_boundary) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
        // H[i+1][0] = 0 with length 0, biased for row i+1; left of a window: -inf
        const int32_t h0 = (int32_t)(((uint32_t)(-dp.mismatch * (i + 1)) << kScoreShift) | (uint32_t)(kLenBias - (i + 1)));
        H[i] = true_boundary ? h0 : kNegE;
        E[i] = kNegE;
    }
}

// best over the last row: strict > keeps the smallest column on ties (policy P3)
__device__ __forceinline__ void dp_track(int32_t hlast, int32_t &best_s, int32_t &best_v) {
    const int32_t s = hlast >> kScoreShift;
    if (s > best_s) { best_s = s; best_v = hlast; }
}

// last row result, then last column rows 1..M-1 (only when the window reaches the read end): a
// last-column cell replaces only when strictly greater, smallest row on ties
template <int M>
__device__ __forceinline__ void dp_result(const int32_t (&H)[M], int32_t best_s, int32_t best_v, bool last_col, const DpParams &dp,
                                          int &score, int &len) {
    int bscore = best_s + dp.mismatch * M;
    int blen = (int)((uint32_t)best_v & kLenMask) + M - kLenBias;
    if (last_col) {
#pragma unroll
        for (int i = 0; i < M - 1; ++i) {
            const int si = (H[i] >> kScoreShift) + dp.mismatch * (i + 1);
            if (si > bscore) { bscore = si; blen = (int)((uint32_t)H[i] & kLenMask) + (i + 1) - kLenBias; }
        }
    }
    score = bscore;
    len = blen;
}

// full-width DP: every column of the read, 16 bases a dword
template <int M>
__device__ __forceinline__ void dp_align(const uint32_t *__restrict__ row, int n, const DpParams &dp, const DpLds &L, int &score, int &len) {
    int32_t HA[M], EA[M];
    dp_init<M>(HA, EA, dp, true);
    int32_t best_s = INT32_MIN, best_v = 0, hl;
    const int nwords = (n + 15) >> 4;
    uint32_t word_next = nwords > 0 ? row[0] : 0u;
    for (int w = 0; w < nwords; ++w) {
        uint32_t word = word_next;
        if (w + 1 < nwords) word_next = row[w + 1];
        const int kmax = min(16, n - 16 * w);
        for (int k = 0; k < kmax; ++k) {                 // in place: a cell is read (hc, ec) before it is written
            dp_column<M>(HA, EA, HA, EA, L.prof + (word & 3u) * kProfStride, dp, hl);
            dp_track(hl, best_s, best_v);
            word >>= 2;
        }
    }
    if (nwords == 0) { score = INT32_MIN / 2; len = 0; return; }
    dp_result<M>(HA, best_s, best_v, true, dp, score, len);
}

// outcome of one alignment task -> spans (+ suffix task for an accepted prefix).  Stores only, one 16-bit half each: the
// prefix pass owns .x, the suffix pass .y (and the prefix pass takes a pending .y back when its alignment fails - whether
// the suffix was pending is in the task's record, so nothing is read: rounds 1-3 read and re-wrote the whole span by index)
__device__ __forceinline__ bool dp_commit(bool valid, const RecHead &h, bool ok, int len, const DpParams &dp, short2 *__restrict__ spans) {
    bool pushS = false;
    if (valid) {
        short *sp = reinterpret_cast<short *>(spans + h.r);
        if (dp.is_prefix) {
            sp[0] = ok ? (short)len : kNone;
            pushS = !dp.independent && ok && h.suf_pending;
            if (!dp.independent && !ok && h.suf_pending) sp[1] = kNone;
        } else {
            sp[1] = (ok && len <= h.n) ? (short)(h.n - len) : kNone;
        }
    }
    return pushS;
}

// ---- dense consumption of a segmented list ---------------------------------------------------------
// Producers write segment b at list[b*seg ..) with cnt[b] entries.  A consumer turns the segments
// into one flat index space with an exclusive prefix sum of cnt[] kept in LDS and finds a task's
// segment by binary search, so its waves are full however unevenly the segments are filled (a
// per-segment consumer ran mostly-empty waves whenever segments held only a few tasks).
struct FlatList {
    uint32_t pre[kMaxGrid + 1];
    uint32_t scan[kBlock];
};
__device__ __forceinline__ void flat_build(FlatList &f, const uint32_t *__restrict__ cnt, uint32_t G) {
    constexpr int PER = kMaxGrid / kBlock;      // 8
    uint32_t loc[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t idx = PER * threadIdx.x + k;
        const uint32_t v = idx < G ? cnt[idx] : 0u;
        loc[k] = sum;
        sum += v;
    }
    f.scan[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
        const uint32_t v = (int)threadIdx.x >= off ? f.scan[threadIdx.x - off] : 0u;
        __syncthreads();
        f.scan[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t excl = f.scan[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) f.pre[PER * threadIdx.x + k] = excl + loc[k];
    if (threadIdx.x == kBlock - 1) f.pre[kMaxGrid] = f.scan[kBlock - 1];
    __syncthreads();
}
// segment holding flat task t (t < total): largest s with pre[s] <= t and pre[s+1] > t
__device__ __forceinline__ uint32_t flat_segment(const FlatList &f, uint32_t t) {
    uint32_t lo = 0, hi = kMaxGrid;             // pre[lo] <= t < pre[hi]
#pragma unroll
    for (int it = 0; it < 11; ++it) {
        const uint32_t mid = (lo + hi) >> 1;
        if (f.pre[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// append read `r` to the suffix list segment of its own scan workgroup (capacity: a read enters listS
// at most once, by k_scan or by the prefix pass); lanes of a wave mostly share the segment
__device__ __forceinline__ void push_suffix(const BatchLists &bl, uint32_t seg, uint32_t r, bool want) {
    uint64_t pending = __ballot(want);
    while (pending) {
        const uint32_t lead = __ffsll((unsigned long long)pending) - 1;
        const uint32_t sseg = __shfl(seg, lead);
        const uint64_t grp = __ballot(want && seg == sseg) & pending;
        uint32_t base = 0;
        if (lane_id() == lead) base = atomicAdd(&bl.cntS[sseg], (uint32_t)__popcll(grp));
        base = __shfl(base, lead);
        if (want && seg == sseg && ((grp >> lane_id()) & 1ull))
            bl.listS[(uint64_t)sseg * bl.seg + base + __popcll(grp & ((1ull << lane_id()) - 1ull))] = r;
        pending &= ~grp;
    }
}

// Full-width DP over the tasks of the segmented list (list, cnt)
template <int M>
__device__ __forceinline__ void dp_full_body(FlatList &fl, DpLds &L, const TaskSrc &src,
                                             const DpParams &dp, short2 *__restrict__ spans, const BatchLists &bl,
                                             const uint32_t *__restrict__ list, const uint32_t *__restrict__ cnt,
                                             uint32_t count_stat, uint64_t *__restrict__ counters, uint32_t bid, uint32_t nb) {
    dp_lds_fill(L, dp);
    flat_build(fl, cnt, bl.grid);
    const uint32_t total = fl.pre[kMaxGrid];
    const uint32_t n_round = (total + WAVE - 1) / WAVE * WAVE;
    for (uint32_t t = bid * kBlock + threadIdx.x; t < n_round; t += nb * kBlock) {
        const bool valid = t < total;
        uint32_t ref = 0, seg = 0;
        RecHead h{0u, 0, false};
        if (valid) {
            seg = flat_segment(fl, t);
            ref = list[(uint64_t)seg * bl.seg + (t - fl.pre[seg])];
            h = task_head(src, ref);
        }
        int score, len;
        dp_align<M>(task_row(src, ref), h.n, dp, L, score, len);
        const bool pushS = dp_commit(valid, h, score >= dp.T, len, dp, spans);
        if (dp.is_prefix) push_suffix(bl, seg, ref, pushS);
    }
    if (bid == 0 && threadIdx.x == 0 && count_stat && total)
        atomicAdd((unsigned long long *)&counters[dp.is_prefix ? C_DP_PREFIX : C_DP_SUFFIX], (unsigned long long)total);
}

template <int M>
__global__ __launch_bounds__(kBlock) void k_dp(int W,
                                               DpParams dp, short2 *__restrict__ spans, BatchLists bl,
                                               const uint32_t *__restrict__ list, const uint32_t *__restrict__ cnt,
                                               uint32_t count_stat, uint64_t *__restrict__ counters) {
    __shared__ FlatList fl;
    __shared__ DpLds L;
    dp_full_body<M>(fl, L, TaskSrc{bl.rows, bl.lens, W}, dp, spans, bl, list, cnt, count_stat, counters, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// k_filter: Myers bit-vector approximate matching as an exact pre-DP filter.
//
// Any alignment the reference would accept (score >= T) loses at most match*m - T against a perfect
// match, hence has at most dp.kerr unit edits (host: max_unit_errors), hence its end column j has
// unit edit distance D[m][j] <= kerr (free start in the read, D[i][0] = i; an adapter overhanging
// the read end is covered because D[m][n] <= D[i][n] + (m-i)), and it spans fewer than m + kerr
// columns.  So every cell on an accepting path lies in columns (jmin - m - kerr, jmax], where
// jmin/jmax are the first/last columns with D[m][j] <= kerr.  Cells off every accepting path can be
// lowered to -inf without changing any cell on one (an optimal predecessor of such a cell is itself
// on an accepting path), so the DP restricted to that window returns the same score, end cell and
// length whenever the full DP would accept, and rejects whenever it would reject.
// ------------------------------------------------------------------------------------------------
// The adapter sits in the TOP m bits of the 32-bit state words (so its last row is bit 31) and the
// low 32-m bits of every Eq mask are ones, which keeps those bits of Pv/Mv/Ph/Mh zero: no masking,
// and "shift in 0" at the adapter's first row is the free-start boundary of approximate searching.
// All per-column work is full-rate VALU (and/or/xor/add/not/shift) plus one LDS read of Eq.
// Detections are collected as one "miss" bit per column and turned into at most two clusters of end
// columns; clusters further apart than m + kerr have disjoint windows.
struct Clusters { int ncl, jmin1, jmax1, jmin2, jmax2, dmin; };   // dmin: smallest D[m][j] over columns 1..4*floor(n/4)

// one Myers column; sc tracks the last adapter row (bit 31) through the horizontal deltas.  Shifts by a
// constant, and/or/xor/add and v_bitop3 on VGPRs are all full rate; carry-chain adds are not.
__device__ __forceinline__ void myers_column(uint32_t Eq, uint32_t &Pv, uint32_t &Mv, int &sc) {
    const uint32_t Xv = Eq | Mv;
    const uint32_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    uint32_t Ph = Mv | ~(Xh | Pv);
    uint32_t Mh = Pv & Xh;
    sc += (int)(Ph >> 31) - (int)(Mh >> 31);
    Ph <<= 1;
    Mh <<= 1;
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
}

// s_eq2[16]: entry (b0 | b1<<2) = {Eq[b0], Eq[b1]}: one 8-byte LDS read serves two columns.
// Every lane of the wave runs the same number of 16-column words (the longest read's); columns past a
// lane's own read are computed on the row's zero padding and masked out of the detections, so the
// column loop has no divergent control flow at all.  The row is held in registers ten words (one
// 40-byte chunk) at a time and the next chunk is fetched while the current one is scanned: a load per
// word inside the loop made every word wait for global memory (59 % of the kernel's wave cycles).
// first 10-word chunk of a task's row (zeros past the read's own words): the caller may fetch it a trip ahead
__device__ __forceinline__ void myers_first_chunk(const uint32_t *__restrict__ row, int n, uint32_t (&xn)[10]) {
    const int nwords = (n + 15) >> 4;
    const uint2 *p = reinterpret_cast<const uint2 *>(row);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        uint2 v = make_uint2(0u, 0u);
        if (2 * i < nwords) v = p[i];
        xn[2 * i] = v.x; xn[2 * i + 1] = v.y;
    }
}
template <int W>
__device__ __forceinline__ void myers_scan(const uint32_t *__restrict__ row, int n, const DpParams &dp, const uint2 *s_eq2,
                                           Clusters &cl, uint32_t (&xn)[10]) {
    constexpr int CW = 10;                             // words per chunk (xn holds chunk 0 on entry)
    uint32_t Pv = dp.pv_init, Mv = 0u;
    int sc = (int)dp.m - dp.kerr - 1;                  // unit edit distance minus (kerr + 1): negative <=> detected
    const int gap = (int)dp.m + dp.kerr;
    cl.ncl = 0; cl.jmin1 = 0; cl.jmax1 = 0; cl.jmin2 = 0; cl.jmax2 = 0;
    // running minimum of sc over every column computed, and its value at the last 4-column boundary inside the
    // lane's own read: columns past the read end are computed on zero padding and must not count
    int smin = sc, smin_own = sc;
    const int nwords = (n + 15) >> 4;
    int nw_max = nwords;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) nw_max = max(nw_max, __shfl_xor(nw_max, o));
    nw_max = __builtin_amdgcn_readfirstlane(nw_max);
    auto eq_pair = [&](uint32_t byte_off) {            // byte_off = entry index * 8
        return *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(s_eq2) + byte_off);
    };
    auto add_run = [&](int jf, int jl) {               // detections jf..jl, all within `gap` of each other
        if (cl.ncl == 0) { cl.ncl = 1; cl.jmin1 = jf; cl.jmax1 = jl; }
        else if (cl.ncl == 1) { if (jf - cl.jmax1 > gap) { cl.ncl = 2; cl.jmin2 = jf; cl.jmax2 = jl; } else cl.jmax1 = jl; }
        else if (cl.ncl == 2) { if (jf - cl.jmax2 > gap) cl.ncl = 3; else cl.jmax2 = jl; }
    };
    auto fetch = [&](int c) {                          // chunk c of this lane's row (zeros past its own words)
        const uint2 *p = reinterpret_cast<const uint2 *>(row + c * CW);
#pragma unroll
        for (int i = 0; i < CW / 2; ++i) {
            uint2 v = make_uint2(0u, 0u);
            if (c * CW + 2 * i < nwords) v = p[i];     // rows are W = 10 k words: a pair never straddles the row end
            xn[2 * i] = v.x; xn[2 * i + 1] = v.y;
        }
    };
    for (int c = 0; c * CW < nw_max; ++c) {
        uint32_t x[CW];
#pragma unroll
        for (int i = 0; i < CW; ++i) x[i] = xn[i];
        if ((c + 1) * CW < nw_max && W > CW) fetch(c + 1);
#pragma unroll
        for (int wi = 0; wi < CW; ++wi) {
            const int w = c * CW + wi;
            if (w < nw_max) {                          // wave-uniform
                uint32_t word = x[wi];
                // ties this word's LDS addresses to the state the previous word leaves: without it the compiler
                // computes all 80 addresses of a chunk up front (149 VGPRs, 3 waves per SIMD)
                asm volatile("" : "+v"(word), "+v"(Pv));
                uint32_t det = 0;                      // one bit per column, newest in bit 0
                uint2 e[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) e[q] = eq_pair(q == 0 ? (word << 3) & 0x78u : (word >> (4 * q - 3)) & 0x78u);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    myers_column(e[q].x, Pv, Mv, sc); det = funnel(det, (uint32_t)sc, 31); smin = min(smin, sc);
                    myers_column(e[q].y, Pv, Mv, sc); det = funnel(det, (uint32_t)sc, 31); smin = min(smin, sc);
                    if (q & 1) smin_own = 16 * w + 2 * (q + 1) <= n ? smin : smin_own;
                }
                // bit 15-k <=> column 16w+k+1; only the lane's own columns count
                const int kmax = min(16, n - 16 * w);
                det &= kmax > 0 ? (0xFFFFu & ~((1u << (16 - kmax)) - 1u)) : 0u;
                if (det) {
                    if (gap >= 15) {                   // a word's detections always share a cluster
                        add_run(16 * w + 16 - (31 - __clz((int)det)), 16 * w + 16 - (__ffs((int)det) - 1));
                    } else {
                        while (det) {
                            const int hb = 31 - __clz((int)det);
                            det &= ~(1u << hb);
                            add_run(16 * w + 16 - hb, 16 * w + 16 - hb);
                        }
                    }
                }
            }
        }
    }
    cl.dmin = smin_own + dp.kerr + 1;
}

constexpr int kMaxWindow = 64;   // columns the windowed DP holds in registers (5 dwords incl. phase)
constexpr int kBand = 10;        // diagonals the banded DP (dp_banded) computes for class 0
constexpr int kBandN = 5;        // ... and for the narrow class of the block filter (one diagonal of block copies, +-2 unit edits)
// ... and for class 6, the WIDE band: any task with ONE cluster of detected end columns [jmin, jmax] so narrow that
// jmax - jmin + 2 kerr + 1 <= kBandW.  Every accepting alignment ends in a column of the cluster and has at most kerr unit
// edits, gap bases among them, so all of its cells lie on the diagonals [jmin - m - kerr, jmax - m + kerr]:
// the windowed DP's argument with the window cut down to its diagonals.  No certified column of distance <= 1 is needed (the
// bound is the plain kerr), so a library with 15 % substitutions - three mismatches an adapter, DP on every read - goes
// here instead of into a 64-column window: M x 16 cells instead of M x (m + kerr + spread) ~ M x 33.
// tests/tools/band_model.py::wide_case restates decision and DP on the CPU against the full-matrix DP.
#ifndef VF_BAND_W
#define VF_BAND_W 16          // (A/B: make variant NAME=bw20 DEFS=-DVF_BAND_W=20)
#endif
constexpr int kBandW = VF_BAND_W;
constexpr int kBandBias = 4096;  // class-0 window word = first diagonal + kBandBias
#ifndef VF_HAMMING_CERT
#define VF_HAMMING_CERT 1       // (A/B: make variant NAME=noham DEFS=-DVF_HAMMING_CERT=0)
#endif

// ------------------------------------------------------------------------------------------------
// k_bfilter: the block filter, in front of the Myers pass.
//
// A task that has a column of unit edit distance <= 1 has an optimum S* >= T' = max(T, match*m - c_max), so every optimal
// alignment has at most K = kerr_lo unit edits (host: fill_dp; mismatch, base in a gap, unaligned adapter end base).  The
// adapter holds K + 1 disjoint blocks; an edit spoils one block at most, so every such alignment matches one block base
// for base, without a gap inside: an exact copy of the block in the read, at read position q = d + blk_off on some
// diagonal d = j - i.  With at most K gap edits in all, every cell of that alignment lies on diagonals [d - K, d + K].
// So: find EVERY copy of every block (XOR-accumulate over the block's bases at every start position, the machinery of
// k_scan's stage 1; shifts are by the base's index inside its block, so all blocks share them), take the hull
// [dmin, dmax] of their diagonals; if the adapter at dmin or dmax differs from the read in at most one base (that
// certifies the column of distance <= 1) the cells of all optimal alignments lie on [dmin - K, dmax + K]: a band of
// dmax - dmin + 2K + 1 diagonals, class N when that is at most kBandN, class 0 (dp_banded at kBand) when at most kBand.
// Everything else -- no copy at all (adapter-free reads, more than K edits), copies too far apart, a copy whose adapter
// would overhang either read end, a single indel (no diagonal certifies) -- goes to listR for the Myers filter.
// Exactness is the windowed DP's argument with T' in the place of T; tests/tools/band_model.py restates
// the decision and the banded DP on the CPU against the full-matrix DP.
// ------------------------------------------------------------------------------------------------
// One task through the block filter.  x = the task's row (zero beyond the read), rowcol = a lane-private LDS column of
// W + 2 words (stride kBlock): the certification reads three words of the row at a dynamic offset, which registers cannot
// give and a global re-load pays a memory round trip for.  Result: toN / to0 (narrow / class-0 band, first diagonal in
// win), neither = the Myers pass decides.
template <int W, int NB>
__device__ __forceinline__ void bf_classify(const uint32_t (&x)[W + 2], int n, bool valid, const DpParams &dp, uint32_t *rowcol,
                                            bool &toN, bool &to0, uint32_t &win) {
    const int m = (int)dp.m, K = dp.kerr_lo;
    uint32_t maxlen = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) maxlen = max(maxlen, (uint32_t)b < dp.nblk ? dp.blk_len[b] : 0u);
    // every copy of every block: acc slot p of block b stays 0 iff read[p + k] == block[k] for all k
    uint32_t acc[NB][W];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int w = 0; w < W; ++w) acc[b][w] = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if ((uint32_t)k < maxlen) {               // wave-uniform
            uint32_t xs[W];
#pragma unroll
            for (int w = 0; w < W; ++w) xs[w] = k ? funnel(x[w + 1], x[w], 2 * k) : x[w];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if ((uint32_t)b < dp.nblk && (uint32_t)k < dp.blk_len[b]) {
                    const uint32_t bp = to_vgpr_u((uint32_t)((dp.pat >> (2u * (dp.blk_off[b] + (uint32_t)k))) & 3ull) * 0x55555555u);
#pragma unroll
                    for (int w = 0; w < W; ++w) acc[b][w] |= xs[w] ^ bp;
                }
            }
        }
    }
    // copies as one bit per start position, moved from read positions to diagonals (q - blk_off) and united
    uint32_t U[W];
#pragma unroll
    for (int w = 0; w < W; ++w) U[w] = 0;
    uint32_t neg = 0;                             // a copy whose adapter would start left of the read
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if ((uint32_t)b < dp.nblk) {              // wave-uniform
            uint32_t h[W + 2];
#pragma unroll
            for (int w = 0; w < W; ++w) h[w] = ~(acc[b][w] | (acc[b][w] >> 1)) & 0x55555555u;
            h[W] = 0; h[W + 1] = 0;
            const uint32_t off = dp.blk_off[b];   // < 32
            const uint32_t lo_bits = 2u * min(off, 16u), hi_bits = off > 16u ? 2u * (off - 16u) : 0u;
            neg |= h[0] & (lo_bits >= 32u ? 0xFFFFFFFFu : ((1u << lo_bits) - 1u));
            neg |= h[1] & ((1u << hi_bits) - 1u);
            const uint32_t bs = 2u * (off & 15u);
            uint32_t sh[W + 1];
#pragma unroll
            for (int w = 0; w <= W; ++w) sh[w] = funnel(h[w + 1], h[w], bs);
            if (off >= 16u) {                     // wave-uniform
#pragma unroll
                for (int w = 0; w < W; ++w) U[w] |= sh[w + 1];
            } else {
#pragma unroll
                for (int w = 0; w < W; ++w) U[w] |= sh[w];
            }
        }
    }
    uint32_t dfirst = 0xFFFFFFFFu;
    int dlast = -1;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        dfirst = min(dfirst, (((uint32_t)(__ffs((int)U[w]) - 1)) >> 1) + 16u * (uint32_t)w);
        dlast = max(dlast, U[w] ? ((31 - __clz((int)U[w])) >> 1) + 16 * w : -1);
    }
    const bool any = valid && dlast >= 0;
    const int dmin = any ? (int)dfirst : 0, dmax = any ? dlast : 0;
    const bool inside = any && neg == 0u && dmax + m <= n;
    // the column of unit edit distance <= 1: the adapter on diagonal dmin or dmax, at most one base different
    bool cert = false;
    if (__any(inside)) {
        const uint64_t patmask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
#pragma unroll
        for (int w = 0; w < W + 2; ++w) rowcol[w * kBlock] = x[w];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int d = inside ? (e == 0 ? dmin : dmax) : 0;
            const int idx = d >> 4;                       // <= W - 1 (d + m <= n <= 16 W)
            const uint32_t bsh = 2u * (uint32_t)(d & 15);
            const uint32_t d0 = rowcol[idx * kBlock], d1 = rowcol[(idx + 1) * kBlock], d2 = rowcol[(idx + 2) * kBlock];
            const uint32_t lo = funnel(d1, d0, bsh), hi = funnel(d2, d1, bsh);
            uint64_t diff = ((((uint64_t)hi << 32) | lo) ^ dp.pat) & patmask;
            diff = (diff | (diff >> 1)) & 0x5555555555555555ull;
            cert = cert || (inside && __popcll(diff) <= 1);
        }
    }
    const int width = dmax - dmin + 2 * K + 1;
    toN = cert && width <= kBandN;
    to0 = cert && !toN && width <= kBand;
    win = (uint32_t)(dmin - K + kBandBias);
}

// Task classes: tasks of similar trip count share waves (a wave runs its longest window).
//   0 banded (dp_banded), 1..2 one window by length, 3 two disjoint windows, 4 anything else -> full-width DP
__device__ __forceinline__ int window_class(const Clusters &cl, int a1, int a2, int base, int kerr) {
    const int w1 = cl.jmax1 - a1, w2 = cl.jmax2 - a2;
    if (cl.ncl == 1) {
        if (w1 <= min(base + max(8, 2 * kerr - 4), kMaxWindow)) return 1;      // the usual width of a task with two unit edits
        return w1 <= kMaxWindow ? 2 : 4;
    }
    if (cl.ncl == 2 && w1 <= kMaxWindow && w2 <= kMaxWindow) return 3;
    return 4;
}

// class 6 (wide band) for a single-cluster task that is not class 0: first diagonal in av
__device__ __forceinline__ bool wide_band_class(const Clusters &cl, const DpParams &dp, int &av) {
    if (cl.ncl != 1 || dp.kerr < 0 || dp.kerr > kBandW || (cl.jmax1 - cl.jmin1) + 2 * dp.kerr + 1 > kBandW) return false;
    av = cl.jmin1 - (int)dp.m - dp.kerr;
    return true;
}

// ------------------------------------------------------------------------------------------------
// The Hamming certificate (round 4): classes 0 / 6 for single-cluster tasks with TWO OR MORE substitutions.
// Lay the adapter on the cluster's centre diagonal(s) - end column jc = (jmin + jmax) / 2, read bases [jc - m, jc) - and count
// the bases that differ: h.  That ungapped alignment scores match*m - h*(match - mismatch), so the task's optimum is
// S* >= T'' = max(T, that), and every OPTIMAL alignment has at most kq = max_unit_errors(T'') unit edits (host: fill_dp,
// dp.kerr_h[h]; kq >= h, so the certified column itself has D <= h <= kq).  The rest is the class-0 argument with kq in the
// place of kerr_lo: end columns of optimal alignments have D <= kq, D moves by at most one per column, so
// they lie kerr - kq columns inside the cluster [jmin, jmax] of columns with D <= kerr on every side the read does not cut
// off: [lo, hi]; alignments with at most kq gap or overhang bases ending there stay on the diagonals
// [lo - m - kq, hi - m + kq].  A copy with two substitutions (defaults, m = 20: T'' = 50, kq = 3, cluster 7 columns wide)
// gets 2 + 6 + 1 = 9 diagonals - class 0 - where the plain bound kerr = 5 gave 6 + 10 + 1 = 17 > kBandW and the task took a
// 64-column window (546 k of the 1.72 M prefix tasks per 2 M reads of the cfg-4 shape, 70 % of that pass's DP cells).
// tests/tools/band_model.py::hamming_case restates decision and band on the CPU against the full-matrix DP.
// ------------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void hamming_band_class(const uint32_t *__restrict__ row, int n, const Clusters &cl, const DpParams &dp, bool cand,
                                                   int &cls, int &av) {
    if (!__any(cand)) return;
    const int m = (int)dp.m;
    const uint64_t patmask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
    int hmin = 64;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int jc = (cl.jmin1 + cl.jmax1 + e) >> 1;
        const int d = jc - m;
        const bool fits = cand && d >= 0 && jc <= n;           // the whole adapter lies inside the read on this diagonal
        const int idx = fits ? d >> 4 : 0;
        const uint32_t bsh = 2u * (uint32_t)((fits ? d : 0) & 15);
        uint32_t d0 = 0, d1 = 0, d2 = 0;
        if (fits) {
            d0 = row[idx];
            d1 = idx + 1 < W ? row[idx + 1] : 0u;
            d2 = idx + 2 < W ? row[idx + 2] : 0u;
        }
        const uint32_t lo = funnel(d1, d0, bsh), hi = funnel(d2, d1, bsh);
        uint64_t diff = ((((uint64_t)hi << 32) | lo) ^ dp.pat) & patmask;
        diff = (diff | (diff >> 1)) & 0x5555555555555555ull;
        const int h = fits ? (int)__popcll(diff) : 64;
        hmin = min(hmin, h);
    }
    int kq = -1;
#pragma unroll
    for (int h = 1; h < 8; ++h) kq = hmin == h ? dp.kerr_h[h] : kq;
    if (!cand || kq < 1 || kq >= dp.kerr) return;
    const int dl = dp.kerr - kq;
    const int lo = cl.jmin1 > 1 ? cl.jmin1 + dl : 1;
    const int hi = cl.jmax1 < n ? cl.jmax1 - dl : n;
    if (hi < lo) return;
    const int width = hi - lo + 2 * kq + 1;
    if (width <= kBand) { cls = 0; av = lo - m - kq; }
    else if (width <= kBandW) { cls = 6; av = lo - m - kq; }
}

// What the Myers pass makes of one task's clusters: cls 0 banded (kBand), 1..3 windowed, 4 whole read, 6 wide band (kBandW);
// av = first diagonal of a banded class, a1 / a2 = window starts.  (One place for k_filter and k_bmfilter's Myers stage.)
template <int W>
__device__ __forceinline__ void myers_classify(const uint32_t *__restrict__ row, int n, bool any, const Clusters &cl, const DpParams &dp,
                                               int &cls, int &av, int &a1, int &a2) {
    const int base = (int)dp.m + dp.kerr;
    a1 = max(0, cl.jmin1 - base); a2 = max(0, cl.jmin2 - base);
    cls = window_class(cl, a1, a2, base, dp.kerr);
    av = 0;
    // A task whose best own column has unit edit distance <= 1 has an optimum of at least match*m - c_max,
    // so all its optimal alignments have at most kerr_lo unit edits (host: fill_dp).  Their end columns
    // have D <= kerr_lo; D changes by at most one per column, so such a column lies kerr - kerr_lo columns
    // inside the cluster of columns with D <= kerr on every side the read does not cut off.  The cells on
    // those alignments lie on the diagonals j - i in [lo - m - kerr_lo, hi - m + kerr_lo]: class 0 when
    // that band is at most kBand wide.
    if (dp.kerr_lo >= 0 && cl.dmin <= 1 && cl.ncl == 1) {
        const int dl = dp.kerr - dp.kerr_lo;
        const int lo = cl.jmin1 > 1 ? cl.jmin1 + dl : 1;
        const int hi = cl.jmax1 < n ? cl.jmax1 - dl : n;
        if (hi >= lo && hi - lo + 2 * dp.kerr_lo + 1 <= kBand) { cls = 0; av = lo - (int)dp.m - dp.kerr_lo; }
    }
    if (VF_HAMMING_CERT) hamming_band_class<W>(row, n, cl, dp, any && cls != 0 && cl.ncl == 1 && dp.kerr_h[0] != 0, cls, av);
    if (cls != 0 && cls != 6 && wide_band_class(cl, dp, av)) cls = 6;
}

// the Myers pass alone, over every task of the adapter's list (scoring schemes that give the block filter no bound, or
// VFIND_BLOCK_FILTER=0)
template <int W>
__global__ __launch_bounds__(kBlock) void k_filter(DpParams dp, short2 *__restrict__ spans, BatchLists bl,
                                                   uint64_t *__restrict__ counters) {
    const TaskSrc src{bl.rows, bl.lens, W};
    __shared__ uint32_t s_c[6];
    __shared__ uint2 s_eq2[16];
    if (threadIdx.x < 16) s_eq2[threadIdx.x] = make_uint2(dp.eq_top[threadIdx.x & 3], dp.eq_top[threadIdx.x >> 2]);
    const uint32_t *list = dp.is_prefix ? bl.listP : bl.listS;
    const uint32_t *cnt = dp.is_prefix ? bl.cntP : bl.cntS;
    uint32_t stat = 0;
    for (uint32_t b = blockIdx.x; b < bl.grid; b += gridDim.x) {
        const uint32_t ntask = cnt[b];
        if (threadIdx.x < 6) s_c[threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t n_round = (ntask + WAVE - 1) / WAVE * WAVE;
        const uint64_t seg0 = (uint64_t)b * bl.seg;
        for (uint32_t t = threadIdx.x; t < n_round; t += kBlock) {
            const bool valid = t < ntask;
            uint32_t r = 0;                              // (a task entry from here on)
            int n = 0;
            if (valid) { r = list[seg0 + t]; n = (int)src.lens[r & kTaskIndexMask]; }
            Clusters cl;
            uint32_t x0[10];
            myers_first_chunk(task_row(src, r), n, x0);
            myers_scan<W>(task_row(src, r), n, dp, s_eq2, cl, x0);
            const bool any = valid && cl.ncl > 0;
            if (valid && !any) (void)dp_commit(true, task_head(src, r), false, 0, dp, spans);      // (the header again: nothing of it is kept across the scan)
            int cls, av, a1, a2;
            myers_classify<W>(task_row(src, r), n, any, cl, dp, cls, av, a1, a2);
            {
                const bool toB = any && cls == 6;
                const uint32_t ib = wave_append(&s_c[5], toB);
                if (toB) { bl.listB[seg0 + ib] = r; bl.winB[seg0 + ib] = (uint32_t)(av + kBandBias); }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool mine = any && cls == q;
                const uint32_t iw = wave_append(&s_c[q], mine);
                if (mine) {
                    bl.listW[q][seg0 + iw] = r;
                    bl.winW[q][seg0 + iw] = q == 0 ? (uint32_t)(av + kBandBias) : ((uint32_t)a1 | ((uint32_t)cl.jmax1 << 16));
                    if (q == 3) bl.winW2[seg0 + iw] = (uint32_t)a2 | ((uint32_t)cl.jmax2 << 16);
                }
            }
            const bool toF = any && cls == 4;
            const uint32_t jf = wave_append(&s_c[4], toF);
            if (toF) bl.listF[seg0 + jf] = r;
        }
        __syncthreads();
        if (threadIdx.x < 4) bl.cntW[threadIdx.x][b] = s_c[threadIdx.x];
        if (threadIdx.x == 4) bl.cntF[b] = s_c[4];
        if (threadIdx.x == 6) { bl.cntB[b] = s_c[5]; if (s_c[5]) atomicAdd(&bl.tot[(dp.is_prefix ? 0u : 8u) + 6u], s_c[5]); }
        if (threadIdx.x < 5 && s_c[threadIdx.x]) atomicAdd(&bl.tot[(dp.is_prefix ? 0u : 8u) + threadIdx.x], s_c[threadIdx.x]);
        if (threadIdx.x == 5) bl.cntN[b] = 0;
        stat += ntask;
        __syncthreads();
    }
    if (threadIdx.x == 0 && stat)
        atomicAdd((unsigned long long *)&counters[dp.is_prefix ? C_DP_PREFIX : C_DP_SUFFIX], (unsigned long long)stat);
}

// ------------------------------------------------------------------------------------------------
// Stages of k_bmfilter (block filter, then Myers pass over what it left), written as per-workgroup loops over a task
// segment that fetch the task of the next trip while they work on the current one (task index two trips ahead, row one
// trip ahead).
// Round 2 tried three other shapes of this stage, all bit-identical in their results, none faster where it counts (cfg 2, two
// batch lanes; DESIGN.md section 8 has the numbers): two per-segment launches (block filter 51 us + Myers pass of the rest
// 52 us on one lane, against 115 us for this kernel, but 0.98 against 0.91 ms per step next to the other lane's k_finish);
// k_front, a persistent kernel with scan, both block filters and the prefix Myers pass, every list workgroup-local (448 us
// against 439 us for its parts: all workgroups walk through the stages in step, nothing overlaps; 1.04-1.15 ms per step);
// and two FLAT launches (every workgroup an equal share of all tasks through a prefix sum over the segments: 50 + 69 us on
// one lane, 1.02-1.06 ms per step).
// ------------------------------------------------------------------------------------------------
#ifndef VF_BF_GIVE_UP_BELOW
#define VF_BF_GIVE_UP_BELOW 2     // fifths of a wave's first trip the block filter must classify for the wave to go on with it (0 = always goes on)
#endif
// trips of one wave over a task list of the workgroup: lane l of wave w takes tasks w*64 + l + 256 i
struct TripIter {
    uint32_t t, ntask;
    __device__ __forceinline__ TripIter(uint32_t ntask_) : t(threadIdx.x), ntask(ntask_) {}
    __device__ __forceinline__ uint32_t trips() const {       // wave-uniform
        const uint32_t wbase = threadIdx.x & ~(uint32_t)(WAVE - 1);
        return ntask > wbase ? (ntask - wbase + kBlock - 1) / kBlock : 0u;
    }
};

template <int W, int NB>
__device__ __forceinline__ void front_bf_stage(const TaskSrc &src, const DpParams &dp,
                                               const uint32_t *list, uint32_t ntask, const ClassSet &cs, uint64_t seg0, uint32_t *s_c,
                                               uint32_t *rowcol) {
    // the task index is fetched a trip ahead, the row is not: a second row in registers (12 more, 106 in all) cost the
    // kernel two of its six waves per SIMD and made it slower than the plain loop
    TripIter it(ntask);
    const uint32_t ntrip = it.trips();
    uint32_t t = it.t;
    bool v_n = t < ntask;
    uint32_t r_n = v_n ? list[seg0 + t] : 0u;
    // A wave whose FIRST trip's tasks the block filter mostly cannot classify (a library with many substitutions: at 15 % only
    // one adapter copy in six has at most one) hands the rest of its tasks to the Myers pass unseen: the block filter's trip -
    // a row gather and ~800 instructions - only pays when it spares most tasks the Myers pass and gives them the narrow band.
    // Any class is exact, so the results do not depend on who classifies (VF_BF_GIVE_UP_BELOW: classified share x 5 of the first trip).
    bool give_up = false;                             // (wave-uniform)
    for (uint32_t i = 0; i < ntrip; ++i, t += kBlock) {
        const uint32_t r = r_n;
        const bool valid = v_n;
        if (give_up) {
            v_n = t + kBlock < ntask;
            r_n = v_n ? list[seg0 + t + kBlock] : 0u;
            const uint32_t iR = wave_append(&s_c[2], valid);
            if (valid) cs.listR[seg0 + iR] = r;
            continue;
        }
        uint32_t x[W + 2];
        int n = 0;
        if (valid) { n = (int)src.lens[r & kTaskIndexMask]; load_row<W>(src.rows, r & kTaskIndexMask, x); }      // (r: a task entry)
        else {
#pragma unroll
            for (int k = 0; k < W + 2; ++k) x[k] = 0;
        }
        v_n = t + kBlock < ntask;
        r_n = v_n ? list[seg0 + t + kBlock] : 0u;
        bool toN, to0;
        uint32_t win;
        bf_classify<W, NB>(x, n, valid, dp, rowcol, toN, to0, win);
        const bool toR = valid && !toN && !to0;
        const uint32_t iN = wave_append(&s_c[0], toN);
        if (toN) { cs.listN[seg0 + iN] = r; cs.winN[seg0 + iN] = win; }
        const uint32_t i0 = wave_append(&s_c[1], to0);
        if (to0) { cs.listW[0][seg0 + i0] = r; cs.winW[0][seg0 + i0] = win; }
        const uint32_t iR = wave_append(&s_c[2], toR);
        if (toR) cs.listR[seg0 + iR] = r;
        if (i == 0 && VF_BF_GIVE_UP_BELOW > 0) {
            const uint32_t seen = (uint32_t)__popcll(__ballot(valid)), hit = (uint32_t)__popcll(__ballot(toN || to0));
            give_up = seen >= 32u && hit * 5u < seen * (uint32_t)VF_BF_GIVE_UP_BELOW;
        }
    }
}

// Myers filter over the nR tasks a block filter left in cs.listR of this workgroup's segment: classes 0..4 (s_my counts them,
// s_my[0] starts at the number of class-0 tasks the block filter put into the segment) or "no alignment can pass".
// Task index one trip ahead.
template <int W>
__device__ __forceinline__ void front_myers_stage(const TaskSrc &src, const DpParams &dp,
                                                  short2 *__restrict__ spans, const ClassSet &cs, uint32_t nR, uint64_t seg0,
                                                  uint32_t *s_my, const uint2 *s_eq2) {
    TripIter it(nR);
    const uint32_t ntrip = it.trips();
    uint32_t t = it.t;
    bool v_n = t < nR;
    uint32_t r_n = v_n ? cs.listR[seg0 + t] : 0u;
    for (uint32_t i = 0; i < ntrip; ++i, t += kBlock) {
        const uint32_t r = r_n;                          // (a task entry)
        const bool valid = v_n;
        const int n = valid ? (int)src.lens[r & kTaskIndexMask] : 0;
        uint32_t x0[10];
        myers_first_chunk(task_row(src, r), n, x0);
        v_n = t + kBlock < nR;
        r_n = v_n ? cs.listR[seg0 + t + kBlock] : 0u;
        Clusters cl;
        myers_scan<W>(task_row(src, r), n, dp, s_eq2, cl, x0);
        const bool any = valid && cl.ncl > 0;
        if (valid && !any) (void)dp_commit(true, task_head(src, r), false, 0, dp, spans);      // (the header again: nothing of it is kept across the scan)
        int cls, av, a1, a2;
        myers_classify<W>(task_row(src, r), n, any, cl, dp, cls, av, a1, a2);
        {
            const bool toB = any && cls == 6;
            const uint32_t ib = wave_append(&s_my[5], toB);
            if (toB) { cs.listB[seg0 + ib] = r; cs.winB[seg0 + ib] = (uint32_t)(av + kBandBias); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool mine = any && cls == q;
            const uint32_t iw = wave_append(&s_my[q], mine);
            if (mine) {
                cs.listW[q][seg0 + iw] = r;
                cs.winW[q][seg0 + iw] = q == 0 ? (uint32_t)(av + kBandBias) : ((uint32_t)a1 | ((uint32_t)cl.jmax1 << 16));
                if (q == 3) cs.winW2[seg0 + iw] = (uint32_t)a2 | ((uint32_t)cl.jmax2 << 16);
            }
        }
        const bool toF = any && cls == 4;
        const uint32_t jf = wave_append(&s_my[4], toF);
        if (toF) cs.listF[seg0 + jf] = r;
    }
}

// k_bmfilter: block filter and Myers pass of ONE adapter in one launch: a workgroup takes a
// task segment through the block filter, then its rest through the Myers filter -- one launch and one tail per adapter
// instead of two, the rest list never leaves the workgroup.
template <int W, int NB>
__global__ __launch_bounds__(kBlock) void k_bmfilter(DpParams dp,
                                                     short2 *__restrict__ spans, BatchLists bl, uint64_t *__restrict__ counters) {
    const TaskSrc src{bl.rows, bl.lens, W};
    __shared__ uint32_t s_bf[3];
    __shared__ uint32_t s_my[6];
    __shared__ uint2 s_eq2[16];
    __shared__ uint32_t s_row[(W + 2) * kBlock];
    if (threadIdx.x < 16) s_eq2[threadIdx.x] = make_uint2(dp.eq_top[threadIdx.x & 3], dp.eq_top[threadIdx.x >> 2]);
    const uint32_t *list = dp.is_prefix ? bl.listP : bl.listS;
    const uint32_t *cnt = dp.is_prefix ? bl.cntP : bl.cntS;
    ClassSet cs;
    cs.listN = bl.listN; cs.winN = bl.winN; cs.winW2 = bl.winW2; cs.listF = bl.listF; cs.listR = bl.listR;
    cs.listB = bl.listB; cs.winB = bl.winB;
#pragma unroll
    for (int q = 0; q < 4; ++q) { cs.listW[q] = bl.listW[q]; cs.winW[q] = bl.winW[q]; }
    uint32_t stat = 0;
    for (uint32_t sb = blockIdx.x; sb < bl.grid; sb += gridDim.x) {
        const uint32_t ntask = cnt[sb];
        if (threadIdx.x < 3) s_bf[threadIdx.x] = 0;
        if (threadIdx.x < 6) s_my[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t seg0 = (uint64_t)sb * bl.seg;
        if (ntask) front_bf_stage<W, NB>(src, dp, list, ntask, cs, seg0, s_bf, s_row + threadIdx.x);
        __syncthreads();
        const uint32_t nR = s_bf[2];
        if (threadIdx.x == 0) s_my[0] = s_bf[1];
        __syncthreads();
        if (nR) front_myers_stage<W>(src, dp, spans, cs, nR, seg0, s_my, s_eq2);
        __syncthreads();
        if (threadIdx.x == 0) {
            bl.cntN[sb] = s_bf[0]; bl.cntR[sb] = nR;
            bl.cntW[0][sb] = s_my[0]; bl.cntW[1][sb] = s_my[1]; bl.cntW[2][sb] = s_my[2]; bl.cntW[3][sb] = s_my[3]; bl.cntF[sb] = s_my[4];
            const uint32_t tb = dp.is_prefix ? 0u : 8u;
            if (s_bf[0]) atomicAdd(&bl.tot[tb + 5u], s_bf[0]);
            for (int q = 0; q < 5; ++q) if (s_my[q]) atomicAdd(&bl.tot[tb + (uint32_t)q], s_my[q]);
            bl.cntB[sb] = s_my[5];
            if (s_my[5]) atomicAdd(&bl.tot[tb + 6u], s_my[5]);
        }
        stat += ntask;
        __syncthreads();
    }
    if (threadIdx.x == 0 && stat)
        atomicAdd((unsigned long long *)&counters[dp.is_prefix ? C_DP_PREFIX : C_DP_SUFFIX], (unsigned long long)stat);
}

// the same DP as dp_align over read columns (a, b] only; column a is the boundary: the true first
// column when a == 0, otherwise -inf (every cell left of the window is off all accepting paths)
template <int M>
__device__ __forceinline__ void dp_window_run(int32_t (&HA)[M], int32_t (&EA)[M],
                                              const uint32_t *__restrict__ row, int W, int a, int b, const DpParams &dp,
                                              const DpLds &L, int32_t &best_s, int32_t &best_v) {
    dp_init<M>(HA, EA, dp, a == 0);
    // 5 dwords hold the window (phase <= 15 bases + 64 columns = 79 <= 80 bases)
    const int wi = a >> 4;
    uint32_t q0 = wi + 0 < W ? row[wi + 0] : 0u;
    uint32_t q1 = wi + 1 < W ? row[wi + 1] : 0u;
    uint32_t q2 = wi + 2 < W ? row[wi + 2] : 0u;
    uint32_t q3 = wi + 3 < W ? row[wi + 3] : 0u;
    uint32_t q4 = wi + 4 < W ? row[wi + 4] : 0u;
    const uint32_t ph = 2u * (uint32_t)(a & 15);
    q0 = funnel(q1, q0, ph); q1 = funnel(q2, q1, ph); q2 = funnel(q3, q2, ph); q3 = funnel(q4, q3, ph); q4 >>= ph;
    int32_t hl;
    const int wlen = b - a;
    // in place (a cell is read before it is written): half the registers of a ping-pong pair of columns, which is
    // what lets the DP waves of one batch lane sit next to the k_finish waves of the other
    int t = 0;
    for (; t + 2 <= wlen; t += 2) {
        dp_column<M>(HA, EA, HA, EA, L.prof + (q0 & 3u) * kProfStride, dp, hl);
        dp_track(hl, best_s, best_v);
        asm volatile("" : "+v"(q0) : "v"(HA[M / 2]));        // the second column's profile is fetched while the first one's lower half runs, not before it (registers)
        dp_column<M>(HA, EA, HA, EA, L.prof + ((q0 >> 2) & 3u) * kProfStride, dp, hl);
        dp_track(hl, best_s, best_v);
        q0 = funnel(q1, q0, 4); q1 = funnel(q2, q1, 4); q2 = funnel(q3, q2, 4); q3 = funnel(q4, q3, 4); q4 >>= 4;
    }
    if (t < wlen) {
        dp_column<M>(HA, EA, HA, EA, L.prof + (q0 & 3u) * kProfStride, dp, hl);
        dp_track(hl, best_s, best_v);
    }
}

// one window, or two disjoint windows left to right (the cells between them are off all accepting
// paths, so the second window restarts from a -inf boundary; the last-row best keeps running, the
// last-column cells count only if the final window ends at the read end)
template <int M, bool TWO>
__device__ __forceinline__ void dp_window(const uint32_t *__restrict__ row, int W, int n, int a, int b, int a2, int b2,
                                          const DpParams &dp, const DpLds &L, int &score, int &len) {
    int32_t HA[M], EA[M];
    int32_t best_s = INT32_MIN, best_v = 0;
    if (b - a <= 0) { score = INT32_MIN / 2; len = 0; return; }
    dp_window_run<M>(HA, EA, row, W, a, b, dp, L, best_s, best_v);
    int lastb = b;
    if (TWO) {
        dp_window_run<M>(HA, EA, row, W, a2, b2, dp, L, best_s, best_v);
        lastb = b2;
    }
    dp_result<M>(HA, best_s, best_v, lastb == n, dp, score, len);
}

template <int M, bool TWO>
__device__ __forceinline__ void dpw_body(FlatList &fl, DpLds &L, const TaskSrc &src, int W,
                                         const DpParams &dp, short2 *__restrict__ spans, const BatchLists &bl, int cls, uint32_t bid, uint32_t nb) {
    dp_lds_fill(L, dp);
    const uint32_t *list = bl.listW[cls], *win = bl.winW[cls];
    flat_build(fl, bl.cntW[cls], bl.grid);
    const uint32_t total = fl.pre[kMaxGrid];
    const uint32_t n_round = (total + WAVE - 1) / WAVE * WAVE;
    for (uint32_t t = bid * kBlock + threadIdx.x; t < n_round; t += nb * kBlock) {
        const bool valid = t < total;
        uint32_t r = 0, seg = 0;                         // (r: a record index)
        RecHead h{0u, 0, false};
        int wa = 0, wb = 0, wa2 = 0, wb2 = 0;
        if (valid) {
            seg = flat_segment(fl, t);
            const uint64_t at = (uint64_t)seg * bl.seg + (t - fl.pre[seg]);
            r = list[at];
            h = task_head(src, r);
            const uint32_t w = win[at];
            wa = (int)(w & 0xFFFFu);
            wb = (int)(w >> 16);
            if (TWO) {
                const uint32_t w2 = bl.winW2[at];
                wa2 = (int)(w2 & 0xFFFFu);
                wb2 = (int)(w2 >> 16);
            }
        }
        int score, len;
        dp_window<M, TWO>(task_row(src, r), W, h.n, wa, wb, wa2, wb2, dp, L, score, len);
        const bool pushS = dp_commit(valid, h, score >= dp.T, len, dp, spans);
        if (dp.is_prefix) push_suffix(bl, seg, r, pushS);
    }
}

// ------------------------------------------------------------------------------------------------
// dp_banded<M, B>: the same DP over a band of B diagonals (B = kBand for class 0 of the Myers filter, kBandN for the narrow
// class of the block filter: every optimal alignment of the task stays on diagonals j - i in [av, av + B)).
// Relative column k = j - av runs from 1 to M + B - 1 and holds the rows (k - B, k]: the code is fully unrolled over k, so
// the row a register holds is static, a cell costs what it costs in dp_column and only M * B cells are computed
// instead of M * (m + kerr + spread).  Everything outside the band reads as -inf (entering row, row above
// the band) or as the true boundary value (row 0; column 0 when the band starts left of the read, where
// av < 0 stands for an adapter overhanging the read start): a value that is never above the true one and
// exact on every cell of an optimal alignment, which is all the result depends on.  A lane is active
// from its first real column (j >= 1) to its last (j <= n): per-column exec masks, no divergent loops.
// ------------------------------------------------------------------------------------------------
template <int M, int B>
__device__ __forceinline__ void dp_banded(const uint32_t *__restrict__ row, int W, int n, int av, const DpParams &dp, const DpLds &L,
                                          int &score, int &len) {
    constexpr int KT = M + B - 1;
    int32_t H[M], E[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {                       // column 0 of the read (only read where it is the true boundary)
        H[i] = (int32_t)(((uint32_t)(-dp.mismatch * (i + 1)) << kScoreShift) | (uint32_t)(kLenBias - (i + 1)));
        E[i] = kNegE;
    }
    const int k0 = av < 0 ? 1 - av : 1;
    const int kend = min(KT, n - av);
    // bases of relative columns 1..KT at bits [2(k-1), 2k) of Q: the row shifted by av bases (either way)
    const int wb = av >> 4;                             // floor
    const uint32_t ph = 2u * (uint32_t)(av & 15);
    uint32_t x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = (wb + t >= 0 && wb + t < W) ? row[wb + t] : 0u;
    uint32_t Q[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) Q[t] = funnel(x[t + 1], x[t], ph);
    const int32_t e_open = to_vgpr(dp.e_open_k), e_ext = to_vgpr(dp.e_ext_k), f_open = to_vgpr(dp.f_open_k), f_ext = to_vgpr(dp.f_ext_k);
    int32_t best_s = -(1 << 24), best_v = 0;
    bool done = false;
    score = 0; len = 0;
#pragma unroll
    for (int k = 1; k <= KT; ++k) {
        if (k >= k0 && k <= kend) {
            const int r0 = k - B + 1 > 1 ? k - B + 1 : 1, r1 = k < M ? k : M;     // rows of this column, 1-based
            const uint32_t b = (Q[(k - 1) >> 4] >> (2 * ((k - 1) & 15))) & 3u;
            const int32_t *pc = L.prof + b * kProfStride;
            int32_t pv[M + 3];
#pragma unroll
            for (int q = (r0 - 1) / 4; q <= (r1 - 1) / 4; ++q) {
                const int4 v = reinterpret_cast<const int4 *>(pc)[q];
                pv[4 * q] = v.x; pv[4 * q + 1] = v.y; pv[4 * q + 2] = v.z; pv[4 * q + 3] = v.w;
            }
            int32_t hup = r0 == 1 ? kV0 : kNegE, f1 = kNegF;
            int32_t hdiag = r0 == 1 ? kV0 : H[r0 - 2];
#pragma unroll
            for (int r = r0; r <= r1; ++r) {
                const int32_t hc = r == k ? kNegE : H[r - 1], ec = r == k ? kNegE : E[r - 1];   // row k enters the band here
                const int32_t e = max(hc + e_open, ec + e_ext) & kClr;
                const int32_t f = max(hup + f_open, f1 + f_ext) | (int32_t)(1u << kPrefShift);
                const int32_t d = hdiag + pv[r - 1];
                const int32_t h = max(max(d, f), e) & kClr;
                hdiag = hc;
                H[r - 1] = h;
                E[r - 1] = e;
                hup = h;
                f1 = f;
            }
            if (r1 == M) dp_track(H[M - 1], best_s, best_v);
            if (av + k == n) {                          // the read's last column: policy P3 over the rows of the band
                int bscore = best_s + dp.mismatch * M;
                int blen = (int)((uint32_t)best_v & kLenMask) + M - kLenBias;
#pragma unroll
                for (int r = r0; r <= (r1 < M - 1 ? r1 : M - 1); ++r) {
                    const int si = (H[r - 1] >> kScoreShift) + dp.mismatch * r;
                    if (si > bscore) { bscore = si; blen = (int)((uint32_t)H[r - 1] & kLenMask) + r - kLenBias; }
                }
                score = bscore; len = blen; done = true;
            }
        }
    }
    if (!done) {
        score = best_s + dp.mismatch * M;
        len = (int)((uint32_t)best_v & kLenMask) + M - kLenBias;
    }
}

template <int M, int B>
__device__ __forceinline__ void dpb_body(FlatList &fl, DpLds &L, const TaskSrc &src, int W,
                                         const DpParams &dp, short2 *__restrict__ spans, const BatchLists &bl,
                                         const uint32_t *__restrict__ list, const uint32_t *__restrict__ win, const uint32_t *__restrict__ cnt,
                                         uint32_t bid, uint32_t nb) {
    dp_lds_fill(L, dp);
    flat_build(fl, cnt, bl.grid);
    const uint32_t total = fl.pre[kMaxGrid];
    const uint32_t n_round = (total + WAVE - 1) / WAVE * WAVE;
    for (uint32_t t = bid * kBlock + threadIdx.x; t < n_round; t += nb * kBlock) {
        const bool valid = t < total;
        uint32_t r = 0, seg = 0;                         // (r: a record index)
        RecHead h{0u, 0, false};
        int av = 0;
        if (valid) {
            seg = flat_segment(fl, t);
            const uint64_t at = (uint64_t)seg * bl.seg + (t - fl.pre[seg]);
            r = list[at];
            h = task_head(src, r);
            av = (int)win[at] - kBandBias;
        }
        int score, len;
        dp_banded<M, B>(task_row(src, r), W, h.n, av, dp, L, score, len);
        const bool pushS = dp_commit(valid, h, score >= dp.T, len, dp, spans);
        if (dp.is_prefix) push_suffix(bl, seg, r, pushS);
    }
}

// ------------------------------------------------------------------------------------------------
// dp_coop: the DP of ONE task by one whole wave -- an adapter row per lane, anti-diagonal by anti-diagonal -- for the rare
// classes whose single-lane DP is the launch's tail: a whole read (class 4) costs one lane 150 x 20 cells = 33 000
// instructions, 70 us with the wave alone on its SIMD, and k_dpall could not end before it; two windows (class 3) 30-40 us.
// Here lane i holds row i + 1; at step t it computes the cell of column j = t - i from its own previous cell (west), the
// cell lane i - 1 computed a step ago (north: one cross-lane move) and the one it computed two steps ago (north-west,
// kept from the previous step): n + M - 1 steps of about 35 instructions.  Plain int32 score / length pairs and the tie
// rules written out as in k_generic's g_sg (open only if strictly greater; diagonal, then north, then west; last row
// strictly greater from the left, last-column rows strictly greater from the top) -- policy P, no packed cell.
// Columns (a, b] of the read: a == 0 is the true first column, otherwise everything left of a is off all accepting paths
// (-inf); for two windows the hull (a1, b2] is computed -- cells between the windows that the windowed DP lowers to -inf
// get some value not above their true one here, which changes no cell on an accepting path.
// ------------------------------------------------------------------------------------------------
template <int M>
__device__ __forceinline__ void dp_coop(const uint32_t *__restrict__ row, int W, int n, int a, int b, const DpParams &dp, uint32_t *s_row,
                                        int &score, int &len) {
    const int lane = (int)lane_id();
    constexpr int NEG = INT32_MIN / 4;
    if (lane < W) s_row[lane] = row[lane];                   // W <= 40: the task's row, wave-private LDS
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool rowlane = lane < M;
    const uint32_t qc = rowlane ? (uint32_t)(dp.pat >> (2 * lane)) & 3u : 0u;
    const int match = dp.match, mismatch = dp.mismatch, gopen = dp.gopen, gext = dp.gext;
    int h = a == 0 ? 0 : NEG, hl = 0, e = NEG, el = 0, f = NEG, fl = 0;      // this row's cell at column a
    int dg_h = 0, dg_hl = 0;                                  // H[i-1][j-1]: row 0 for lane 0, column a for the others at their first step
    if (lane > 0) dg_h = a == 0 ? 0 : NEG;
    int best = NEG, bestl = 0, colh = NEG, colhl = 0;
    const int cols = b - a;
    for (int t = 1; t <= cols + M - 1; ++t) {
        int uh = __shfl_up(h, 1), uhl = __shfl_up(hl, 1), uf = __shfl_up(f, 1), ufl = __shfl_up(fl, 1);
        if (lane == 0) { uh = 0; uhl = 0; uf = NEG; ufl = 0; }       // row 0: free start anywhere in the read
        const int jr = t - lane;                                  // column relative to a, 1-based
        if (rowlane && jr >= 1 && jr <= cols) {
            const int j = a + jr;
            const uint32_t wd = s_row[(j - 1) >> 4];
            const uint32_t rc = (wd >> (2 * ((j - 1) & 15))) & 3u;
            const int eo = h - gopen, ee = e - gext;
            int ne, nel;
            if (eo > ee) { ne = eo; nel = hl + 1; } else { ne = ee; nel = el + 1; }
            const int fo = uh - gopen, fe = uf - gext;
            int nf, nfl;
            if (fo > fe) { nf = fo; nfl = uhl + 1; } else { nf = fe; nfl = ufl + 1; }
            const int d = dg_h + (qc == rc ? match : mismatch), dl = dg_hl + 1;
            const int nh = max(d, max(ne, nf));
            const int nhl = nh == d ? dl : (nh == nf ? nfl : nel);
            h = nh; hl = nhl; e = ne; el = nel; f = nf; fl = nfl;
            if (lane == M - 1 && nh > best) { best = nh; bestl = nhl; }
            if (j == n) { colh = nh; colhl = nhl; }
        }
        dg_h = uh; dg_hl = uhl;
    }
    int bs = __shfl(best, M - 1), bl_ = __shfl(bestl, M - 1);
    if (b == n) {
#pragma unroll 1
        for (int i = 0; i < M - 1; ++i) {
            const int ci = __shfl(colh, i), cl = __shfl(colhl, i);
            if (ci > bs) { bs = ci; bl_ = cl; }
        }
    }
    score = bs; len = bl_;
}

// classes 3 (two windows: their hull) and 4 (whole read), one task per wave
template <int M>
__device__ __forceinline__ void dpc_body(FlatList &fl, const TaskSrc &src, int W,
                                         const DpParams &dp, short2 *__restrict__ spans, const BatchLists &bl, bool two_windows,
                                         uint32_t *s_rows, uint32_t bid, uint32_t nb) {
    const uint32_t *list = two_windows ? bl.listW[3] : bl.listF;
    flat_build(fl, two_windows ? bl.cntW[3] : bl.cntF, bl.grid);
    const uint32_t total = fl.pre[kMaxGrid];
    const uint32_t wave = threadIdx.x / WAVE, nwaves = kBlock / WAVE;
    uint32_t *s_row = s_rows + wave * 40;
    for (uint32_t t = bid * nwaves + wave; t < total; t += nb * nwaves) {          // wave-uniform
        const uint32_t seg = flat_segment(fl, t);
        const uint64_t at = (uint64_t)seg * bl.seg + (t - fl.pre[seg]);
        const uint32_t r = list[at];                     // (a task entry)
        const RecHead h = task_head(src, r);
        const int n = h.n;
        int a = 0, b = n;
        if (two_windows) { a = (int)(bl.winW[3][at] & 0xFFFFu); b = (int)(bl.winW2[at] >> 16); }
        int score = INT32_MIN / 2, len = 0;
        if (b - a > 0) dp_coop<M>(task_row(src, r), W, n, a, b, dp, s_row, score, len);
        const bool lead = lane_id() == 0;
        const bool pushS = dp_commit(lead, h, score >= dp.T, len, dp, spans);
        if (dp.is_prefix) push_suffix(bl, seg, r, pushS);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// k_dpall: every task class of one filter pass in ONE launch.  The filter leaves the number of tasks per class
// in bl.tot; every workgroup derives the same split of the grid from them -- blocks in proportion to tasks x
// cells per task, at least one block for a class that has tasks, no more blocks than the class has waves --
// and works on its class.  The rare classes (long windows, two windows, full-width rest) are a handful of tasks
// each and cost one wave's latency (a whole read for the full-width rest): as launches of their own, on a side
// stream, they took longer than the two big classes next to them, and the side streams shared hardware queues
// with the table clear (a lane stalled behind another lane's k_finish: profiles/, checkpoint M).
// ------------------------------------------------------------------------------------------------
#ifdef VF_FINISH_ABLATION
// diagnosis build: per workgroup of k_dpall {class, start, prologue done, end} in 10 ns ticks (s_memrealtime); [0] prefix, [1] suffix pass
__device__ uint64_t g_dp_trace[2][kMaxGrid][4];
#define VF_DP_STAMP(slot) do { if (threadIdx.x == 0) g_dp_trace[dp.is_prefix ? 0 : 1][blockIdx.x][slot] = wall_clock64(); } while (0)
#define VF_DP_CLASS(c) do { if (threadIdx.x == 0) g_dp_trace[dp.is_prefix ? 0 : 1][blockIdx.x][0] = (uint64_t)(c); } while (0)
#else
#define VF_DP_STAMP(slot) do {} while (0)
#define VF_DP_CLASS(c) do {} while (0)
#endif
template <int M>
__global__ __launch_bounds__(kBlock) void k_dpall(int W,
                                                  DpParams dp, short2 *__restrict__ spans, BatchLists bl, uint64_t *__restrict__ counters) {
    const TaskSrc src{bl.rows, bl.lens, W};
    __shared__ FlatList fl;
    __shared__ DpLds L;
    __shared__ uint32_t s_rows[(kBlock / WAVE) * 40];
    VF_DP_CLASS(99);
    VF_DP_STAMP(1);
    const uint32_t *tot = bl.tot + (dp.is_prefix ? 0u : 8u);
    // relative cost of a task: banded M x kBand cells; one window ~ M x (m + kerr + 8); long / two windows ~ M x 64 x 1..2;
    // full width M x read length
    // index 5 = the narrow band of the block filter (M x kBandN cells).  Classes 3 and 4 (two windows, whole read) are a
    // handful of tasks in a low-error library: then they run one task per WAVE (dp_coop, ~12 us a task instead of 30-70 us for
    // one lane), and a task costs a wave what 64 lanes of the other classes cost it together; as big classes (15 % adapter
    // errors: 75 000 two-window tasks) they stay one task per lane.
    uint32_t cost[7] = {4u, 16u, 32u, 48u, 128u, 2u, (uint32_t)(kBandW * 7 / 16)};                // (index 6 = the wide band: M x kBandW cells)
    uint32_t nbk[7], t5[7];
    uint64_t wsum = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) t5[q] = __builtin_amdgcn_readfirstlane(tot[q]);
    const bool coop3 = t5[3] <= dp.coop_max, coop4 = t5[4] <= dp.coop_max;
    if (coop3) cost[3] = 600u;
    if (coop4) cost[4] = 600u;
#pragma unroll
    for (int q = 0; q < 7; ++q) wsum += (uint64_t)t5[q] * cost[q];
    if (wsum == 0) return;
    const uint32_t G = gridDim.x;
    uint32_t used = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        uint32_t b = t5[q] ? (uint32_t)((uint64_t)G * t5[q] * cost[q] / wsum) : 0u;
        // one workgroup per 256 tasks at most; classes 3 and 4 run one task per WAVE (dp_coop): one workgroup per 4 tasks
        const uint32_t per_wg = ((q == 3 && coop3) || (q == 4 && coop4)) ? (uint32_t)(kBlock / WAVE) : (uint32_t)kBlock;
        const uint32_t cap = (t5[q] + per_wg - 1) / per_wg;
        if (t5[q] && b < 1) b = 1;
        if (q > 0 && q < 5 && cap <= 2 * b + 8) b = cap;                   // a small class runs as one trip: its latency is the launch's tail
        if (b > cap) b = cap;
        nbk[q] = b;
        used += b;
    }
    // the minimum of one block and the single-trip raise can exceed the grid: the largest classes pay (a class keeps one block)
#pragma unroll
    for (int it = 0; it < 7; ++it) {
        if (used > G) {
            int big = 0;
#pragma unroll
            for (int q = 1; q < 7; ++q) if (nbk[q] > nbk[big]) big = q;
            const uint32_t cut = min(used - G, nbk[big] - 1u);
            nbk[big] -= cut;
            used -= cut;
        }
    }
    uint32_t b0 = 0;
    const uint32_t b = blockIdx.x;
    VF_DP_STAMP(2);
    if (b < b0 + nbk[5]) { VF_DP_CLASS(5); dpb_body<M, kBandN>(fl, L, src, W, dp, spans, bl, bl.listN, bl.winN, bl.cntN, b - b0, nbk[5]); VF_DP_STAMP(3); return; }
    b0 += nbk[5];
    if (b < b0 + nbk[0]) { VF_DP_CLASS(0); dpb_body<M, kBand>(fl, L, src, W, dp, spans, bl, bl.listW[0], bl.winW[0], bl.cntW[0], b - b0, nbk[0]); VF_DP_STAMP(3); return; }
    b0 += nbk[0];
    if (b < b0 + nbk[6]) { VF_DP_CLASS(6); dpb_body<M, kBandW>(fl, L, src, W, dp, spans, bl, bl.listB, bl.winB, bl.cntB, b - b0, nbk[6]); VF_DP_STAMP(3); return; }
    b0 += nbk[6];
    if (b < b0 + nbk[1]) { VF_DP_CLASS(1); dpw_body<M, false>(fl, L, src, W, dp, spans, bl, 1, b - b0, nbk[1]); VF_DP_STAMP(3); return; }
    b0 += nbk[1];
    if (b < b0 + nbk[2]) { VF_DP_CLASS(2); dpw_body<M, false>(fl, L, src, W, dp, spans, bl, 2, b - b0, nbk[2]); VF_DP_STAMP(3); return; }
    b0 += nbk[2];
    if (b < b0 + nbk[3]) {
        VF_DP_CLASS(3);
        if (coop3) dpc_body<M>(fl, src, W, dp, spans, bl, true, s_rows, b - b0, nbk[3]);
        else dpw_body<M, true>(fl, L, src, W, dp, spans, bl, 3, b - b0, nbk[3]);
        VF_DP_STAMP(3);
        return;
    }
    b0 += nbk[3];
    if (b < b0 + nbk[4]) {
        VF_DP_CLASS(4);
        if (coop4) dpc_body<M>(fl, src, W, dp, spans, bl, false, s_rows, b - b0, nbk[4]);
        else dp_full_body<M>(fl, L, src, dp, spans, bl, bl.listF, bl.cntF, 0u, counters, b - b0, nbk[4]);
        VF_DP_STAMP(3);
    }
}

// ------------------------------------------------------------------------------------------------
// device count table
// ------------------------------------------------------------------------------------------------
struct Hash { uint32_t h1, h2; };
__device__ __forceinline__ void hash_init(Hash &h, uint32_t meta) {
    h.h1 = 0x811C9DC5u ^ meta;
    h.h2 = 0x9E3779B9u + meta * 0x85EBCA6Bu;
}
__device__ __forceinline__ void hash_step(Hash &h, uint32_t k) {
    h.h1 = (h.h1 ^ k) * 0x01000193u;
    h.h2 = (h.h2 + k) * 0x9E3779B1u;
    h.h2 ^= h.h2 >> 15;
}
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ void hash_fin(Hash &h) { h.h1 = fmix32(h.h1 ^ (h.h2 >> 7)); h.h2 = fmix32(h.h2) | 1u; }

__device__ __forceinline__ uint32_t key_dwords(uint32_t kind, uint32_t len) {
    return kind == kKindPacked ? (len + 15u) >> 4 : (len + 3u) >> 2;
}

constexpr int kInlineDwords = 14;        // key dwords held inside the slot

// key held in registers (static indexing)
template <int ND> struct RegKey {
    static constexpr uint32_t kMaxDwords = (uint32_t)ND;      // (a key held in ND registers cannot be longer: the LONG-key paths compile out)
    const uint32_t (&k)[ND];
    template <class F> __device__ __forceinline__ void each(uint32_t nd, F f) const {
#pragma unroll
        for (int d = 0; d < ND; ++d) if ((uint32_t)d < nd) f((uint32_t)d, k[d]);
    }
    // first kInlineDwords dwords, zero padded (register keys are zero beyond their length)
    __device__ __forceinline__ void head(uint32_t, uint32_t (&o)[kInlineDwords]) const {
#pragma unroll
        for (int d = 0; d < kInlineDwords; ++d) o[d] = d < ND ? k[d < ND ? d : 0] : 0u;
    }
};
// key held in memory, element d at p[d*stride]
struct MemKey {
    static constexpr uint32_t kMaxDwords = 0xFFFFFFFFu;
    const uint32_t *p;
    uint64_t stride;
    uint32_t mask = 0xFFFFFFFFu;             // (a key read out of a tagged slot line: ~kTagBits)
    template <class F> __device__ __forceinline__ void each(uint32_t nd, F f) const {
        for (uint32_t d = 0; d < nd; ++d) f(d, p[d * stride] & mask);
    }
    __device__ __forceinline__ void head(uint32_t nd, uint32_t (&o)[kInlineDwords]) const {
#pragma unroll
        for (int d = 0; d < kInlineDwords; ++d) o[d] = (uint32_t)d < nd ? (p[d * stride] & mask) : 0u;
    }
};

// ---- device count table: one 64-byte slot per key ---------------------------------------------------
//   slot = { u64 w0; u32 key[14] }                       (a lookup touches exactly one 64-byte line)
//   w0   = tag | len<<32 | kind<<48 | LONG<<50 | OVF<<51 | READY<<63 ;  0 = empty
// Keys of up to 14 dwords (56 peptide letters / 224 bases) live in the slot, zero padded, so a probe
// compares all 14 dwords without predication.  Longer keys (LONG) keep their arena offset in key[0..1].
// Counts sit in their own array: atomics execute at the memory side and drop the line from L2, so a
// count sharing the slot's line made every probe of a hot key miss L2 (measured 3x slower).
// Protocol: claim by 64-bit CAS on w0 (READY clear), write the key, s_waitcnt vmcnt(0), store w0|READY.
// Every word another workgroup may read is written and read with agent-scope (sc1) accesses; all of
// them are write-once, so a stale view can only ever cause a duplicate entry, never a false match
// (the host sums duplicate strings when it builds the result).  A negative compare is only believed
// when it comes from a load issued after READY had already been observed.
constexpr uint64_t kW0Ready = 1ull << 63, kW0Long = 1ull << 50, kW0Ovf = 1ull << 51;
// kW0One: the entry's count is one more than its count words say - the read that inserted the key left its own count in
// the claim instead of a count word (an insert then costs no count line: in a cleared table that is one random line read
// and written back less for every new key, 1.6 M of the 8.7 M random lines of a cfg-2 step).  Set only by the claim, so
// nothing ever has to change it; every reader of counts adds it (count_of).
constexpr uint64_t kW0One = 1ull << 52;
// Epoch tags (Table::epoch, vf_device.h): w0 bits 53..56.  A slot whose w0 is 0 or carries another epoch is FREE (claimed by
// a CAS that expects exactly the stale word); a letter key's dwords carry the epoch in their bytes' top bits, so a dword left
// over from an earlier epoch - or never written - reads as "the owner's store has not landed yet", exactly what a zero
// dword means in an untagged table.  Untagged tables (epoch 0: packed-DNA keys have no spare bits) are cleared by zeroing.
constexpr int kW0EpochShift = 53;
constexpr uint64_t kW0EpochMask = 15ull << kW0EpochShift;
__device__ __forceinline__ uint64_t w0_epoch(const Table &t) { return (uint64_t)t.epoch << kW0EpochShift; }
__device__ __forceinline__ bool w0_free(const Table &t, uint64_t w0) { return w0 == 0ull || (w0 & kW0EpochMask) != w0_epoch(t); }
__device__ __forceinline__ bool w0_live(const Table &t, uint64_t w0) { return (w0 & kW0Ready) && !(w0 & kW0Ovf) && (w0 & kW0EpochMask) == w0_epoch(t); }
// has the owner's store of this key dword landed?
__device__ __forceinline__ bool dword_written(const Table &t, uint32_t v) { return t.key_tag ? (v & kTagBits) == t.key_tag : v != 0u; }
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
// A LONG line's key[0..1]: the key's arena offset (36 bits: the arena ends at 2^36 dwords) and, above it, 28 check bits
// made of the line's tag and the table's epoch - never 0.  Insurance: nothing read from a slot line is used as an address
// unless it validates itself and lies inside the arena.  (Round 3's fault was a reader that paired the READY w0 a lost CAS
// had returned with the key words of the load BEFORE the CAS - table_find_or_insert, image_older - and followed the letters
// an earlier epoch had left in key[0..1]; while tables were cleared by zeroing the stale word was 0, the compare against
// the key at arena[0] failed and the reader inserted a duplicate, never noticed.  The hardware keeps its side: a writer's
// store, s_waitcnt, flag store are seen in that order from another XCD - scripts/ubench/visibility.hip, 1.7 x 10^8 watched
// transitions per variant, none torn.)  A reader that finds the word wanting looks again.
constexpr uint64_t kLongOffMask = (1ull << 36) - 1ull;
__device__ __forceinline__ uint32_t long_check(const Table &t, uint32_t tag) { return (((tag >> 4) ^ t.epoch) & 0x07FFFFFFu) | 0x08000000u; }
__device__ __forceinline__ uint64_t long_word(const Table &t, uint32_t tag, uint64_t off) { return off | ((uint64_t)long_check(t, tag) << 36); }
__device__ __forceinline__ bool long_word_ok(const Table &t, uint32_t tag, uint64_t word, uint32_t nd) {
    return (uint32_t)(word >> 36) == long_check(t, tag) && (word & kLongOffMask) + nd <= t.arena_cap;
}
// A volatile access to LDS through a GENERIC pointer compiles to a FLAT instruction (the compiler gives up on the address
// space), and FLAT accesses to LDS travel through the vector-memory path: they are not ordered with the DS instructions of
// the same wave.  k_finish read a line stage it had just written with ds_write through flat_load (27 FLAT instructions in
// the kernel until round 3) - never seen to go wrong, but a stale read there is a wild global store.  Every volatile LDS
// access goes through an explicit LDS pointer: DS instructions, executed in issue order.
#define VF_LDS(T, p) ((volatile __attribute__((address_space(3))) T *)(p))
#ifdef VF_BOUNDS_CHECK      // diagnosis build only (make variant NAME=bc DEFS=-DVF_BOUNDS_CHECK): the first violated bound is
// recorded in the debug counters (code, two values, workgroup) and the access is skipped
__device__ __forceinline__ bool vf_bound(const Table &t, bool ok, uint64_t code, uint64_t v0, uint64_t v1) {
    if (!ok && atomicCAS((unsigned long long *)&t.counters[C_DBG0], 0ull, (unsigned long long)code) == 0ull) {
        t.counters[C_DBG1] = v0; t.counters[C_DBG2] = v1; t.counters[C_DBG3] = blockIdx.x;
    }
    return ok;
}
#define VF_BOUND(t, ok, code, v0, v1) vf_bound(t, ok, code, v0, v1)
#else
#define VF_BOUND(t, ok, code, v0, v1) true
#endif

__device__ __forceinline__ void slot_load(const uint4 *p, uint4 &a, uint4 &b, uint4 &c, uint4 &d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
}
// whole-slot write-through store: four 16-byte sc1 stores (8-byte sc1 stores are one fabric write each
// and cost 2.7x per byte, MI355X_MICROARCH "stores of each flavour")
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void slot_store(uint4 *p, u32x4 a, u32x4 b, u32x4 c, u32x4 d) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\t"
                 "global_store_dwordx4 %0, %2, off offset:16 sc1\n\t"
                 "global_store_dwordx4 %0, %3, off offset:32 sc1\n\t"
                 "global_store_dwordx4 %0, %4, off offset:48 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 :: "v"(p), "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
}
__device__ __forceinline__ void slot_store_nowait(uint4 *p, u32x4 a, u32x4 b, u32x4 c, u32x4 d) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\t"
                 "global_store_dwordx4 %0, %2, off offset:16 sc1\n\t"
                 "global_store_dwordx4 %0, %3, off offset:32 sc1\n\t"
                 "global_store_dwordx4 %0, %4, off offset:48 sc1"
                 :: "v"(p), "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
}
// Counts: one 64-bit word per slot, apart from the slot lines, and k_finish never touches it.  An occurrence of a key
// that is already in the table is written to the batch's count log as the key's slot index (CountLog, vf_device.h) and
// applied after the kernel, range by range (k_log_apply).  Rounds 1-2 counted in place, with an atomic per occurrence
// executed in the issuing XCD's L2 on one count array per XCD: every tail key then cost a count line fetched and written
// back (384 MB of the 975 MB k_finish moved per 10 M reads) and - loads, stores and atomics of a wave retire in issue
// order - every trip's slot lookup waited behind the count atomics of the trip before it.
// A count of any size from anywhere (merge, growth, the byte-wise path): device scope, executed at the memory side.
__device__ __forceinline__ void base_add(const Table &t, uint32_t slot, unsigned long long v) {
    __hip_atomic_fetch_add(t.counts + slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t count_of(const Table &t, uint64_t slot, uint64_t w0) {      // the entry's count (see kW0One)
    return t.counts[slot] + ((w0 & kW0One) ? 1ull : 0ull);
}

// ---- workgroup-local state: arena chunks for LONG keys, insert counters --------------------------
// LONG keys take arena space from a chunk owned by the workgroup; only a refill touches the global
// bump pointer.  The chunk state survives across launches in t.chunk_state[row][block], one row per
// stream, so the arena loses at most one partial chunk per (stream, workgroup) in total.
//   state = (chunk_index + 1) << 32 | dwords_used      (0 = no chunk yet)
struct BlockTable {
    unsigned long long chunk;                // LDS copy of the chunk state
    uint32_t new_entries;                    // keys inserted by this workgroup (flushed at kernel end)
    uint32_t new_key_dwords;                 // their total key dwords
};
__device__ __forceinline__ void block_table_begin(BlockTable &bt, const Table &t) {
    if (threadIdx.x == 0) {
        bt.chunk = t.chunk_state[(uint64_t)t.chunk_row * kMaxGrid + blockIdx.x];
        if (!VF_BOUND(t, (bt.chunk >> 32) <= t.arena_cap / kChunk && (uint32_t)bt.chunk <= (1u << 24), 1, bt.chunk, t.chunk_row)) bt.chunk = 0;
        bt.new_entries = 0;
        bt.new_key_dwords = 0;
    }
}
__device__ __forceinline__ void block_table_end(BlockTable &bt, const Table &t) {     // after a __syncthreads()
    if (threadIdx.x == 0) {
        t.chunk_state[(uint64_t)t.chunk_row * kMaxGrid + blockIdx.x] = bt.chunk;
        if (bt.new_entries) {
            atomicAdd((unsigned long long *)&t.counters[C_ENTRIES], (unsigned long long)bt.new_entries);
            atomicAdd((unsigned long long *)&t.counters[C_KEY_DWORDS], (unsigned long long)bt.new_key_dwords);
        }
    }
}
enum { kAllocRetry = 0, kAllocOk = 1, kAllocFull = 2 };
__device__ __forceinline__ int arena_alloc(BlockTable &bt, const Table &t, uint32_t nd, uint64_t &off) {
    if (nd > kChunk / 4) {                   // oversized key: whole chunks straight from the global pointer
        const uint64_t span = (uint64_t)(nd + kChunk - 1) / kChunk * kChunk;
        off = atomicAdd((unsigned long long *)&t.counters[C_ARENA_TOP], (unsigned long long)span);
        return off + span <= t.arena_cap ? kAllocOk : kAllocFull;
    }
    const unsigned long long old = atomicAdd(&bt.chunk, (unsigned long long)nd);
    const uint32_t used = (uint32_t)old, idxp1 = (uint32_t)(old >> 32);
    if (idxp1 != 0u && used + nd <= kChunk) { off = (uint64_t)(idxp1 - 1u) * kChunk + used; return kAllocOk; }
    const bool crosser = idxp1 == 0u ? used == 0u : used <= kChunk;
    if (!crosser) return kAllocRetry;        // the lane that crossed the chunk end is fetching a new chunk
    const uint64_t base = atomicAdd((unsigned long long *)&t.counters[C_ARENA_TOP], (unsigned long long)kChunk);
    if (base + kChunk > t.arena_cap) {
        atomicExch(&bt.chunk, 0ull);         // let a later lane try again (and fail the same way)
        return kAllocFull;
    }
    atomicExch(&bt.chunk, ((unsigned long long)(base / kChunk + 1u) << 32) | (unsigned long long)nd);
    off = base;
    return kAllocOk;
}

__device__ __forceinline__ void slot_load_key(const uint4 *p, uint32_t nd, bool can, uint32_t tag, uint4 &a, uint4 &b, uint4 &c, uint4 &d);
#ifndef VF_SLOW_PARTIAL
#define VF_SLOW_PARTIAL 1       // (A/B: make variant NAME=slowfull DEFS=-DVF_SLOW_PARTIAL=0: whole slot lines in the insert queue, as until round 4)
#endif
// Finds key's slot, inserting the key if absent; returns kNoSlot on a capacity error.
// have_off: arena offset the (LONG) key already occupies (table growth re-inserts entries whose keys stay where they
// are), kNoOff = the key is copied into a fresh arena allocation.
// one: the caller is counting ONE occurrence; if this lane is the one that inserts the key, the claim carries that count
// (kW0One) and `counted` comes back true - the caller then adds nothing.
constexpr uint64_t kNoOff = ~0ull;
template <class K>
__device__ __forceinline__ uint32_t table_find_or_insert(const Table &t, BlockTable &bt, const K &key, uint32_t kind, uint32_t len, bool active,
                                                         uint64_t have_off, bool one, bool &counted) {
    counted = false;
    uint32_t found = kNoSlot;
    const uint32_t nd = key_dwords(kind, len);
    const uint32_t meta = len | (kind << 16);
    Hash h;
    hash_init(h, meta);
    key.each(nd, [&](uint32_t, uint32_t v) { hash_step(h, v); });
    hash_fin(h);
    const bool is_long = K::kMaxDwords > (uint32_t)kInlineDwords && nd > (uint32_t)kInlineDwords;
    const uint64_t w0mine = (uint64_t)h.h2 | ((uint64_t)meta << 32) | (is_long ? kW0Long : 0ull) | w0_epoch(t);
    const uint64_t w0pub = w0mine | (one ? kW0One : 0ull);                // what this lane writes if it inserts
    // Letter keys (kind 0) validate themselves: every dword of a written key is non-zero, so a slot can
    // be claimed READY at once and its key stored without waiting; a reader that meets a zero dword
    // inside the key's length knows the line is still in flight and looks again.  Packed-DNA keys may
    // hold zero dwords and keep the claim -> store -> wait -> READY order.
    const bool selfval = kind == kKindBytes && !is_long;
    const uint64_t w0claim = selfval ? (w0pub | kW0Ready) : w0pub;
    uint32_t kin[kInlineDwords];
    key.head(nd, kin);
    if (selfval) {                           // the slot's image of a letter key carries the epoch in every dword (padding too)
#pragma unroll
        for (int d = 0; d < kInlineDwords; ++d) kin[d] |= t.key_tag;
    }
    uint32_t slot = h.h1 & t.slot_mask;
    uint32_t probes = 0;
    bool done = !active;
    bool claimed = false;                    // this lane won the slot and must publish its key
    bool confirmed = false;                  // READY already seen at this slot (a mismatch is now final)
    // The loop condition is a wave-wide vote: no lane leaves before every lane is done.  A lane that
    // finds a slot claimed but not yet published simply looks again on the next trip; because the
    // claiming lane (possibly in this very wave) publishes inside the loop body, before the vote,
    // the wait cannot deadlock and the compiler cannot sink the publish past the loop exit.
#ifdef VF_INSERT_COUNTERS
    if (lane_id() == 0) atomicAdd((unsigned long long *)&t.counters[C_DBG0], 1ull);
#endif
    while (__any(!done)) {
#ifdef VF_INSERT_COUNTERS
        if (lane_id() == 0) atomicAdd((unsigned long long *)&t.counters[C_DBG1], 1ull);
#endif
        if (!done) {
            uint4 *sp = t.slots + (uint64_t)slot * 4;
            bool advance = false;
            uint4 l0 = make_uint4(0, 0, 0, 0), l1 = l0, l2 = l0, l3 = l0;
            uint64_t w0 = 0;
            // the w0 a lost CAS returns is NEWER than the line image loaded before it: another lane may have claimed the slot,
            // written its key and set READY in between.  Such a w0 may say "another key" (advance) or "not READY yet" (wait), but
            // the key words that came with the earlier load must not be compared or followed under it (round 3: a LONG key's
            // reader took the letters an earlier epoch had left in key[0..1] for an arena offset).
            bool image_older = false;
            if (!claimed) {
                // Only the 16-byte quarters of the line this lane's key reaches (a LONG key: w0 and the offset word), as in the
                // one-shot lookup: a quarter is one request at the L2, and the queue's loop was 4 of them a trip for every lane
                // in it (round 4's request table: ~5 M of k_finish's 18.8 M read requests a step).  A slot whose w0 matches holds
                // a key of the same length, padded like this one, so the quarters left out compare equal by construction.
                if (VF_SLOW_PARTIAL) slot_load_key(sp, is_long ? 2u : nd, true, selfval ? t.key_tag : 0u, l0, l1, l2, l3);
                else slot_load(sp, l0, l1, l2, l3);
                w0 = (uint64_t)l0.x | ((uint64_t)l0.y << 32);
                if (w0_free(t, w0)) {
                    const uint64_t seen = w0;                    // (0, or a word of an earlier epoch: the CAS expects exactly it)
                    w0 = atomicCAS(reinterpret_cast<unsigned long long *>(sp), (unsigned long long)seen, (unsigned long long)w0claim);
                    claimed = w0 == seen;
                    image_older = !claimed;
#ifdef VF_NO_IMAGE_GUARD        // diagnosis build only (with -DVF_LONG_DIAG): round 3's behaviour, the offset check alone catches it
                    image_older = false;
#endif
#ifdef VF_INSERT_COUNTERS
                    atomicAdd((unsigned long long *)&t.counters[C_DBG2], 1ull);
                    if (claimed) atomicAdd((unsigned long long *)&t.counters[C_DBG3], 1ull);
#endif
                }
            }
            if (claimed) {
                unsigned long long *kw = reinterpret_cast<unsigned long long *>(sp) + 1;
                if (selfval) {
                    slot_store_nowait(sp, u32x4{(uint32_t)w0claim, (uint32_t)(w0claim >> 32), kin[0], kin[1]}, u32x4{kin[2], kin[3], kin[4], kin[5]},
                                      u32x4{kin[6], kin[7], kin[8], kin[9]}, u32x4{kin[10], kin[11], kin[12], kin[13]});
                    atomicAdd(&bt.new_entries, 1u);
                    atomicAdd(&bt.new_key_dwords, nd);
                    found = slot;
                    counted = one;
                    done = true;
                } else if (!is_long) {
                    // the line is ours (w0 claimed, READY clear): rewrite it whole, then publish
                    slot_store(sp, u32x4{(uint32_t)w0pub, (uint32_t)(w0pub >> 32), kin[0], kin[1]}, u32x4{kin[2], kin[3], kin[4], kin[5]},
                               u32x4{kin[6], kin[7], kin[8], kin[9]}, u32x4{kin[10], kin[11], kin[12], kin[13]});
                    st_agent(reinterpret_cast<unsigned long long *>(sp), (unsigned long long)(w0pub | kW0Ready));
                    atomicAdd(&bt.new_entries, 1u);
                    atomicAdd(&bt.new_key_dwords, nd);
                    found = slot;
                    counted = one;
                    done = true;
                } else {
                    uint64_t off = have_off;
                    const int rc = have_off != kNoOff ? (int)kAllocOk : arena_alloc(bt, t, nd, off);
                    if (rc == kAllocOk && !VF_BOUND(t, off + nd <= t.arena_cap, 2, off, nd)) {
                        done = true;
                    } else if (rc == kAllocOk) {
                        if (have_off == kNoOff) key.each(nd, [&](uint32_t d, uint32_t v) { st_agent(&t.arena[off + d], v); });
                        st_agent(kw, (unsigned long long)long_word(t, h.h2, off));
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        st_agent(reinterpret_cast<unsigned long long *>(sp), (unsigned long long)(w0pub | kW0Ready));
                        atomicAdd(&bt.new_entries, 1u);
                        atomicAdd(&bt.new_key_dwords, nd);
                        found = slot;
                        counted = one;
                        done = true;
                    } else if (rc == kAllocFull) {
                        atomicOr((unsigned long long *)&t.counters[C_ERROR], 1ull);
                        st_agent(reinterpret_cast<unsigned long long *>(sp), (unsigned long long)(w0mine | kW0Ready | kW0Ovf));
                        done = true;
                    }
                }
            } else if ((w0 & ~(kW0Ready | kW0Ovf | kW0One)) == w0mine) {
                if (!(w0 & kW0Ready) || image_older) {
                    // claimed by another lane, key not yet published - or published after this trip's load: look at the same
                    // slot again next trip
                } else if (w0 & kW0Ovf) {
                    done = true;
                } else {
                    uint32_t diff = 0;
                    bool offset_landed = true;           // (LONG lines: long_word_ok)
                    if (!is_long) {
                        diff = (l0.z ^ kin[0]) | (l0.w ^ kin[1]) | (l1.x ^ kin[2]) | (l1.y ^ kin[3]) | (l1.z ^ kin[4]) | (l1.w ^ kin[5]) |
                               (l2.x ^ kin[6]) | (l2.y ^ kin[7]) | (l2.z ^ kin[8]) | (l2.w ^ kin[9]) | (l3.x ^ kin[10]) | (l3.y ^ kin[11]) |
                               (l3.z ^ kin[12]) | (l3.w ^ kin[13]);
                    } else {
                        const uint64_t word = (uint64_t)l0.z | ((uint64_t)l0.w << 32);
                        const uint64_t off = word & kLongOffMask;
                        offset_landed = long_word_ok(t, h.h2, word, nd);
#ifdef VF_LONG_DIAG             // diagnosis build only: [C_DBG0] offset words that did not validate, [C_DBG1] LONG lines compared
                        atomicAdd((unsigned long long *)&t.counters[offset_landed ? C_DBG1 : C_DBG0], 1ull);
#endif
                        if (offset_landed) key.each(nd, [&](uint32_t d, uint32_t v) { diff |= ld_agent(&t.arena[off + d]) ^ v; });
                    }
                    if (!offset_landed) {
                        // READY seen, the offset word not yet: the same slot again next trip
                    } else if (diff == 0u) { found = slot; done = true; }
                    else if (selfval) {
                        // a zero dword inside the key's length = the owner's stores are still in flight
                        const uint32_t ln[kInlineDwords] = {l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2.x, l2.y, l2.z, l2.w, l3.x, l3.y, l3.z, l3.w};
                        bool incomplete = false;
#pragma unroll
                        // (in a tagged table the padding dwords carry the epoch too: a stale one means the line is still landing)
                        for (int d = 0; d < kInlineDwords; ++d) incomplete = incomplete || (((uint32_t)d < nd || t.key_tag != 0u) && !dword_written(t, ln[d]));
                        if (!incomplete) advance = true;
                    }
                    else if (!confirmed && !is_long) confirmed = true;      // re-read once now that READY is known
                    else advance = true;
                }
            } else {
                advance = true;
            }
            if (advance) {
                confirmed = false;
                slot = (slot + 1u) & t.slot_mask;
                if (++probes > t.slot_mask) {
                    atomicOr((unsigned long long *)&t.counters[C_ERROR], 2ull);
                    done = true;
                }
            }
        }
    }
    return found;
}

template <class K>
__device__ __forceinline__ void table_add(const Table &t, BlockTable &bt, const K &key, uint32_t kind, uint32_t len, uint64_t cnt, bool active,
                                          uint64_t have_off = kNoOff) {
    bool counted;
    const bool one = cnt == 1ull;
    const uint32_t slot = table_find_or_insert(t, bt, key, kind, len, active, have_off, one, counted);
    if (slot != kNoSlot && !counted) base_add(t, slot, (unsigned long long)cnt);
}

// Workgroup-level count aggregation: a small LDS table keyed by slot index collects the +1s of a
// whole workgroup (all its grid-stride iterations); what it holds at kernel end goes to the count log as
// (slot, count) pairs, so the Zipf head of the variant distribution costs one log entry per workgroup
// instead of one per read.  A slot that loses the LDS entry to another slot is logged occurrence by occurrence.
#ifndef VF_AGG_SIZE
#define VF_AGG_SIZE 1024
#endif
constexpr int kAggSize = VF_AGG_SIZE;
struct Agg {
    uint32_t slot[kAggSize];
    uint32_t cnt[kAggSize];
};
__device__ __forceinline__ void agg_init(Agg &a) {
    for (int i = threadIdx.x; i < kAggSize; i += kBlock) { a.slot[i] = kNoSlot; a.cnt[i] = 0; }
}
// true = counted here
__device__ __forceinline__ bool agg_try(Agg &a, uint32_t slot) {
    const uint32_t i = ((slot * 0x9E3779B1u) >> 16) & (uint32_t)(kAggSize - 1);
    uint32_t cur = *VF_LDS(uint32_t, &a.slot[i]);
    if (cur == kNoSlot) { cur = atomicCAS(&a.slot[i], kNoSlot, slot); if (cur == kNoSlot) cur = slot; }
    if (cur != slot) return false;
    atomicAdd(&a.cnt[i], 1u);
    return true;
}

// ---- the count log as k_finish writes it (CountLog, vf_device.h) ---------------------------------
// Per wave: a segment of slot indices, appended in place with one store per trip - the lanes that log write consecutive
// dwords, the line fills up in L2 over the next trips and leaves it once.  Per workgroup: the (slot, count) pairs of the
// aggregator and the key cache at kernel end, and the number of entries per slot part in LDS (s_tot), added to the
// batch's totals when the workgroup is done: k_log_scatter turns the totals into part starts.
struct WaveLog {
    uint32_t *seg;                           // this wave's segment of log.one
    uint32_t n;                              // entries written (wave-uniform)
};
__device__ __forceinline__ void log_one(WaveLog &wl, const CountLog &log, uint32_t *s_tot, uint32_t slot, bool want) {
    const uint64_t m = __ballot(want);
    if (!m) return;
    if (want) {
        const uint32_t at = wl.n + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
        if (at < log.cap_one) wl.seg[at] = slot;           // (cap_one = the reads a wave can see: never exceeded)
        if (log.parts) atomicAdd(&s_tot[slot >> log.shift], 1u);
    }
    wl.n += (uint32_t)__popcll(m);
}
__device__ __forceinline__ void log_many(const CountLog &log, uint32_t *s_tot, uint32_t *s_nmany, uint32_t slot, uint32_t cnt) {
    const uint32_t at = atomicAdd(s_nmany, 1u);
    if (at < kLogMany) log.many[(uint64_t)blockIdx.x * kLogMany + at] = make_uint2(slot, cnt);
    if (log.parts) atomicAdd(&s_tot[kLogMaxParts + (slot >> log.shift)], 1u);
}
__device__ __forceinline__ void agg_flush(Agg &a, const CountLog &log, uint32_t *s_tot, uint32_t *s_nmany) {
    for (int i = threadIdx.x; i < kAggSize; i += kBlock)
        if (a.slot[i] != kNoSlot && a.cnt[i]) log_many(log, s_tot, s_nmany, a.slot[i], a.cnt[i]);
}

// Workgroup-level key cache: the occurrences of a workgroup's commonest keys, counted in LDS by key, in front of the table.
// The first lane of the workgroup that meets a key while the key's entry is free claims the entry, then takes the table
// path itself (lookup, insert queue) and publishes the slot it ends up with; every later read of that key in the
// workgroup - in the same wave a moment later, or ten thousand reads on - is counted in the entry and asks the table
// nothing.  What it saves is requests to ONE line: the first key of the Zipf stream is 12 % of the reads, the first ten
// a third, and every table lookup of a key is two or three 16-byte requests to the same L2 line from every CU of the
// XCD, served one after the other by that line's channel.  In a cleared table that was the launch's first 100 us: every
// wave's first insert batch held eight lanes of the commonest key and all 3 072 waves ran it at once (rounds 1-2 entered
// a key only once the table had it, i.e. after that storm).  Entry = 16 dwords:
//   { state, h1, h2, meta | slot, count, -, - | key[0..3] | key[4..7] }
//   state 0 free, 1 claimed (key and tags written by the claiming lane right after the claim; slot not yet known),
//   2 slot known.  The last tag word (meta) is release-stored after everything else of the entry and acquire-loaded first by
//   readers (keycache_match), so a reader that sees it sees the whole key; an entry is never replaced; meta is never 0 (a
//   counted key has a length).
// Direct mapped by h1; keys of more than eight dwords are not cached.  Flushed at kernel end as (slot, count) pairs.
#ifndef VF_KEYCACHE_SIZE
#define VF_KEYCACHE_SIZE 256
#endif
constexpr int kKeyCacheSize = VF_KEYCACHE_SIZE;        // power of two
struct KeyCache { u32x4 q[kKeyCacheSize][4]; };
__device__ __forceinline__ void keycache_init(KeyCache &kc) {
    for (int i = threadIdx.x; i < kKeyCacheSize; i += kBlock) {
#pragma unroll
        for (int q = 0; q < 4; ++q) kc.q[i][q] = u32x4{0u, 0u, 0u, 0u};
    }
}
__device__ __forceinline__ uint32_t keycache_index(uint32_t h1) { return ((h1 * 0x9E3779B1u) >> 16) & (uint32_t)(kKeyCacheSize - 1); }
// Publication of an entry's key (C++ / HSA memory model, workgroup scope, LDS): the claiming lane writes the key words, the
// slot word and two of the three tag words with plain stores and then RELEASE-stores the last tag word, `meta` (never 0; 0
// until then - entries are never replaced); a reader ACQUIRE-loads `meta` first and looks at the other words only behind a
// match.  Rounds 2-3 wrote tags after key words with plain volatile stores and leaned on one CU's LDS executing the DS
// operations of different waves in arrival order - true of the hardware, not of the memory model (a false hit would count a
// read against the wrong key's slot, silently).  On gfx950 the release costs one s_waitcnt lgkmcnt(0) in the claiming lane's
// wave, once per entry; the acquire is the wait the compare needs anyway.
__device__ __forceinline__ uint32_t keycache_meta_acquire(KeyCache &kc, uint32_t i) {
    return __hip_atomic_load(reinterpret_cast<uint32_t *>(&kc.q[i][0]) + 3, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool keycache_match(KeyCache &kc, uint32_t i, uint32_t h1, uint32_t h2, uint32_t meta, const uint32_t (&kin)[kInlineDwords]) {
    if (keycache_meta_acquire(kc, i) != meta) return false;
    const u32x4 t0 = *VF_LDS(u32x4, &kc.q[i][0]);
    const u32x4 k0 = *VF_LDS(u32x4, &kc.q[i][2]), k1 = *VF_LDS(u32x4, &kc.q[i][3]);
    const uint32_t diff = (t0.y ^ h1) | (t0.z ^ h2) | (k0.x ^ kin[0]) | (k0.y ^ kin[1]) | (k0.z ^ kin[2]) | (k0.w ^ kin[3]) |
                          (k1.x ^ kin[4]) | (k1.y ^ kin[5]) | (k1.z ^ kin[6]) | (k1.w ^ kin[7]);
    return diff == 0u;
}
// What the cache does with one read of a key: kCacheCounted = counted in the key's entry, done; kCacheOwner = this lane
// has just claimed the entry for the key: it goes on to the table (where its own occurrence is counted like any other) and
// publishes the slot; kCacheNone = the entry belongs to another key (or the key is not cacheable): table path.
enum { kCacheNone = 0, kCacheCounted = 1, kCacheOwner = 2 };
__device__ __forceinline__ int keycache_probe(KeyCache &kc, uint32_t h1, uint32_t h2, uint32_t meta, const uint32_t (&kin)[kInlineDwords], bool can) {
    const uint32_t i = keycache_index(h1);
    uint32_t *st = reinterpret_cast<uint32_t *>(&kc.q[i][0]);
    int role = kCacheNone;
    if (can && *VF_LDS(uint32_t, st) == 0u && atomicCAS(st, 0u, 1u) == 0u) {
        *VF_LDS(u32x4, &kc.q[i][2]) = u32x4{kin[0], kin[1], kin[2], kin[3]};
        *VF_LDS(u32x4, &kc.q[i][3]) = u32x4{kin[4], kin[5], kin[6], kin[7]};
        *VF_LDS(u32x4, &kc.q[i][1]) = u32x4{kNoSlot, 0u, 0u, 0u};
        VF_LDS(uint32_t, st)[1] = h1; VF_LDS(uint32_t, st)[2] = h2;
        __hip_atomic_store(st + 3, meta, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        role = kCacheOwner;
    }
    // (the lanes of this wave that lost the claim to a lane with the same key find the key published: the release store
    // above is earlier in the same instruction stream)
    if (can && role == kCacheNone && keycache_match(kc, i, h1, h2, meta, kin)) {
        atomicAdd(reinterpret_cast<uint32_t *>(&kc.q[i][1]) + 1, 1u);
        role = kCacheCounted;
    }
    return role;
}
// the table slot of a key (found or inserted, by whichever lane of the workgroup got there): noted in the key's entry if
// it has one that still waits for it.  Two lanes may both note a slot - each is a slot of that key (duplicates of a key
// in the table are summed when the result is built), either will do.
__device__ __forceinline__ void keycache_publish(KeyCache &kc, uint32_t h1, uint32_t h2, uint32_t meta, const uint32_t (&kin)[kInlineDwords], uint32_t slot, bool can) {
    if (!can || slot == kNoSlot) return;
    const uint32_t i = keycache_index(h1);
    volatile __attribute__((address_space(3))) uint32_t *st = VF_LDS(uint32_t, &kc.q[i][0]);
    if (st[0] != 1u || !keycache_match(kc, i, h1, h2, meta, kin)) return;
    VF_LDS(uint32_t, &kc.q[i][1])[0] = slot;
    // (slot and state are read back only by keycache_flush, behind the workgroup barrier at kernel end; the release keeps
    // the pair ordered for whoever reads them earlier)
    __hip_atomic_store(reinterpret_cast<uint32_t *>(&kc.q[i][0]), 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// after the workgroup's last table access.  An entry with occurrences but no slot means its owner's insert failed: the
// table or the arena is full, which the insert has flagged; flagged again here so that it cannot pass unseen.
__device__ __forceinline__ void keycache_flush(KeyCache &kc, const Table &t, const CountLog &log, uint32_t *s_tot, uint32_t *s_nmany) {
    for (int i = threadIdx.x; i < kKeyCacheSize; i += kBlock) {
        const u32x4 t0 = kc.q[i][0], t1 = kc.q[i][1];
        if (t0.x == 2u && t1.y && VF_BOUND(t, t1.x <= t.slot_mask, 8, t1.x, t1.y)) log_many(log, s_tot, s_nmany, t1.x, t1.y);
        else if (t0.x == 1u && t1.y) atomicOr((unsigned long long *)&t.counters[C_ERROR], 16ull);
    }
}

// ------------------------------------------------------------------------------------------------
// k_finish: extract seq[start..end], translate, count
// ------------------------------------------------------------------------------------------------
// standard genetic code, index = b0 + 4*b1 + 16*b2 (little-endian codon as it sits in the row)
__constant__ uint8_t c_codon_le[64];

// The key of one read: region [start,end) of its row, as peptide letters (lib.rs:16-44) or, with
// skip_translation, as the 2-bit bases themselves (expanded to letters by the host).
template <int W, bool DNA> struct FinishKey {
    static constexpr int MAXC = (16 * W) / 3;                    // codons a row can hold
    static constexpr int ND = DNA ? W : (MAXC + 3) / 4;          // key dwords
    static constexpr uint32_t kKind = DNA ? kKindPacked : kKindBytes;
};

// x is consumed (shifted in place).  len = key length in letters / bases, act = the read is counted.
template <int W, bool DNA>
__device__ __forceinline__ void finish_key(uint32_t (&x)[W + 2], int start, int end, bool ok, const uint8_t *lut,
                                           uint32_t (&key)[FinishKey<W, DNA>::ND], uint32_t &len, bool &act) {
    // shift the row right by `start` bases: word-granular barrel shifter, then bit funnel
    const uint32_t sh = 2u * (uint32_t)(ok ? start : 0), ws = sh >> 5, bs = sh & 31u;
#pragma unroll
    for (int s = 1; s < W; s <<= 1) {
        const bool take = (ws & (uint32_t)s) != 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint32_t from = w + s < W ? x[w + s] : 0u;
            x[w] = take ? from : x[w];
        }
    }
#pragma unroll
    for (int w = 0; w < W; ++w) x[w] = funnel(x[w + 1], x[w], bs);
    const uint32_t L = ok ? (uint32_t)(end - start) : 0u;
    if constexpr (DNA) {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int rem = (int)L - 16 * w;
            const uint32_t mask = rem >= 16 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (2 * rem)) - 1u));
            key[w] = x[w] & mask;
        }
        len = L;
        act = ok;
    } else {
        constexpr int MAXC = FinishKey<W, DNA>::MAXC;
        const bool tr = ok && (L % 3u) == 0u;                    // lib.rs:17-19
        const uint32_t nc = tr ? L / 3u : 0u;
#pragma unroll
        for (int d = 0; d < FinishKey<W, DNA>::ND; ++d) {
            uint32_t kd = 0;
            if (__any((uint32_t)(4 * d) < nc)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 4 * d + q;
                    if (c < MAXC) {
                        const int bit = 6 * c, wi = bit >> 5, bo = bit & 31;
                        const uint32_t cod = (bo <= 26 ? (x[wi] >> bo) : funnel(x[wi + 1], x[wi], bo)) & 63u;
                        kd |= (uint32_t)lut[cod] << (8 * q);
                    }
                }
                const int rem = (int)nc - 4 * d;
                const uint32_t mask = rem >= 4 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
                kd &= mask;
            }
            key[d] = kd;
        }
        len = nc;
        act = tr;
    }
}

// One-shot lookup of a key that is (most often) already in the table: hash, one slot line, compare.
// Anything else - empty slot, another key in the slot, a key still being published, a LONG key - is a
// miss here and goes to the workgroup's slow list, where table_find_or_insert runs with full waves.
// the home slot's line for the one-shot lookup: only the 16-byte quarters the lane's own key reaches (w0 and two key
// dwords, then four dwords each), and nothing for a lane without a key.  A slot whose w0 matches holds a key of the
// same length, zero padded like the lane's, so the quarters left out compare equal by construction; a quarter is
// one request at the L2, and the lookup is bound by requests (a 12-letter peptide needs two of the four).
__device__ __forceinline__ void slot_load_key(const uint4 *p, uint32_t nd, bool can, uint32_t tag, uint4 &a, uint4 &b, uint4 &c, uint4 &d) {
    // (a quarter that is not fetched compares as the key's padding: zero, or the epoch pattern in a tagged table)
    u32x4 va = {0u, 0u, tag, tag}, vb = {tag, tag, tag, tag}, vc = vb, vd = vb;
    // one asm block, predicated through exec: as separate statements the compiler copied a destination register
    // between the load and the wait (it cannot know the result is still in flight)
    const uint64_t m1 = __ballot(can), m2 = __ballot(can && nd > 2), m3 = __ballot(can && nd > 6), m4 = __ballot(can && nd > 10);
    uint64_t save;
    asm volatile("s_mov_b64 %4, exec\n\t"
                 "s_mov_b64 exec, %6\n\t"
                 "global_load_dwordx4 %0, %5, off sc1\n\t"
                 "s_mov_b64 exec, %7\n\t"
                 "global_load_dwordx4 %1, %5, off offset:16 sc1\n\t"
                 "s_mov_b64 exec, %8\n\t"
                 "global_load_dwordx4 %2, %5, off offset:32 sc1\n\t"
                 "s_mov_b64 exec, %9\n\t"
                 "global_load_dwordx4 %3, %5, off offset:48 sc1\n\t"
                 "s_mov_b64 exec, %4\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "+v"(va), "+v"(vb), "+v"(vc), "+v"(vd), "=&s"(save)
                 : "v"(p), "s"(m1), "s"(m2), "s"(m3), "s"(m4)
                 : "memory");
    a = make_uint4(va.x, va.y, va.z, va.w); b = make_uint4(vb.x, vb.y, vb.z, vb.w);
    c = make_uint4(vc.x, vc.y, vc.z, vc.w); d = make_uint4(vd.x, vd.y, vd.z, vd.w);
}
// The one-shot lookup's line loads, two lanes to a line (round 4; built, measured, OFF).  A lane loading the quarters of its own
// slot line issues one L2 request per 16-byte quarter - 2.25 a lookup for cfg 2's peptides, 11 M of the 20 M L2 requests k_finish
// makes a step.  Here the wave's 64 lookups are served in two rounds of 32: lanes 2k and 2k + 1 load the first and the second
// quarter of lookup 32 r + k - 32 bytes next to each other, ONE request - straight into LDS (global_load_lds_dwordx4:
// destination = M0 + 16 x lane, which is exactly row k of a [32][32 B] image), and every lane then reads its own row back.  The
// third and fourth quarter (keys of more than 24 letters: a minority) come per lane as before.  Bit-identical (79 device tests),
// k_finish's L2 requests 20.1 M -> 15.6 M a step - and the step does not move (cfg 2 0.735-0.745 against 0.737-0.746 ms, same
// box, three rounds; cfg-4 shape +1.5 %, cfg-5 shape -1 %): the request list through LDS, the barrier and the read-back put two
// LDS round trips in front of and behind every trip's one memory round trip, 132 registers instead of 125, 5.6 KB more LDS.
// Not every L2 request costs the same (EXPERIMENTS R4.6).
// req: the wave's {slot, quarters} pairs; img: its [64][32 B] image (2 KiB).  Wave-uniform call.
#ifndef VF_COOP_LOOKUP
#define VF_COOP_LOOKUP 0        // (A/B: make variant NAME=coop DEFS=-DVF_COOP_LOOKUP=1)
#endif
#if VF_COOP_LOOKUP
typedef uint32_t u32x2g __attribute__((ext_vector_type(2)));
struct LookupStage { u32x4 img[WAVE][2]; u32x2g req[WAVE]; };
__device__ __forceinline__ void slot_load_coop(LookupStage &st, const uint4 *slots, uint32_t slot, uint32_t nd, bool can, uint32_t tag,
                                               uint4 &a, uint4 &b, uint4 &c, uint4 &d) {
    const uint32_t lane = lane_id();
    const uint32_t nq = !can ? 0u : (nd > 10u ? 4u : nd > 6u ? 3u : nd > 2u ? 2u : 1u);
    *VF_LDS(u32x2g, &st.req[lane]) = u32x2g{slot, nq};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&st.img[0][0]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const u32x2g rq = *VF_LDS(u32x2g, &st.req[32 * r + (lane >> 1)]);
        const uint32_t h = lane & 1u;
        const bool act = rq.y > h;                                     // the lookup wants this quarter
        const uint4 *src = slots + (uint64_t)rq.x * 4 + h;
        const uint64_t m = __ballot(act);
        uint64_t save_exec;
        uint32_t save_m0;
        asm volatile("s_mov_b64 %0, exec\n\t"
                     "s_mov_b32 %1, m0\n\t"
                     "s_mov_b32 m0, %4\n\t"
                     "s_mov_b64 exec, %3\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, off sc1\n\t"
                     "s_mov_b64 exec, %0\n\t"
                     "s_mov_b32 m0, %1"
                     : "=&s"(save_exec), "=&s"(save_m0)
                     : "v"(src), "s"(m), "s"(lds_base + (uint32_t)r * 1024u)
                     : "memory");
    }
    // quarters three and four: per lane, as slot_load_key does
    u32x4 vc = {tag, tag, tag, tag}, vd = vc;
    {
        const uint4 *p = slots + (uint64_t)slot * 4;
        const uint64_t m3 = __ballot(nq > 2u), m4 = __ballot(nq > 3u);
        uint64_t save;
        asm volatile("s_mov_b64 %2, exec\n\t"
                     "s_mov_b64 exec, %4\n\t"
                     "global_load_dwordx4 %0, %3, off offset:32 sc1\n\t"
                     "s_mov_b64 exec, %5\n\t"
                     "global_load_dwordx4 %1, %3, off offset:48 sc1\n\t"
                     "s_mov_b64 exec, %2\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "+v"(vc), "+v"(vd), "=&s"(save)
                     : "v"(p), "s"(m3), "s"(m4)
                     : "memory");
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (a quarter that was not fetched compares as the key's padding: zero, or the epoch pattern in a tagged table)
    u32x4 va = *VF_LDS(u32x4, &st.img[lane][0]), vb = *VF_LDS(u32x4, &st.img[lane][1]);
    if (nq < 1u) va = u32x4{0u, 0u, tag, tag};
    if (nq < 2u) vb = u32x4{tag, tag, tag, tag};
    a = make_uint4(va.x, va.y, va.z, va.w); b = make_uint4(vb.x, vb.y, vb.z, vb.w);
    c = make_uint4(vc.x, vc.y, vc.z, vc.w); d = make_uint4(vd.x, vd.y, vd.z, vd.w);
    __builtin_amdgcn_wave_barrier();                                   // (the image is free for the next call)
}
#endif
struct FastProbe {
    uint32_t slot;
    uint32_t h1, h2, meta;
    uint64_t w0want;
    uint32_t kin[kInlineDwords];
    uint32_t nd;
    bool can;
};
template <int ND>
__device__ __forceinline__ void fast_prepare(const Table &t, const uint32_t (&key)[ND], uint32_t kind, uint32_t len, bool act, FastProbe &p) {
    const uint32_t nd = key_dwords(kind, len);
    const uint32_t meta = len | (kind << 16);
    Hash h;
    hash_init(h, meta);
    const RegKey<ND> rk{key};
    rk.each(nd, [&](uint32_t, uint32_t v) { hash_step(h, v); });
    hash_fin(h);
    p.can = act && nd <= (uint32_t)kInlineDwords;
    p.nd = nd;
    p.h1 = h.h1; p.h2 = h.h2; p.meta = meta;
    p.w0want = (uint64_t)h.h2 | ((uint64_t)meta << 32) | kW0Ready | w0_epoch(t);
    p.slot = h.h1 & t.slot_mask;
    rk.head(nd, p.kin);
    if (kind == kKindBytes) {                // the slot's image of a letter key: the epoch in every dword, padding too (0 in an untagged table)
#pragma unroll
        for (int d = 0; d < kInlineDwords; ++d) p.kin[d] |= t.key_tag;
    }
}
__device__ __forceinline__ bool fast_check(const FastProbe &p, const uint4 &l0, const uint4 &l1, const uint4 &l2, const uint4 &l3) {
    const uint64_t w0 = (uint64_t)l0.x | ((uint64_t)l0.y << 32);
    const uint32_t diff = (l0.z ^ p.kin[0]) | (l0.w ^ p.kin[1]) | (l1.x ^ p.kin[2]) | (l1.y ^ p.kin[3]) | (l1.z ^ p.kin[4]) | (l1.w ^ p.kin[5]) |
                          (l2.x ^ p.kin[6]) | (l2.y ^ p.kin[7]) | (l2.z ^ p.kin[8]) | (l2.w ^ p.kin[9]) | (l3.x ^ p.kin[10]) | (l3.y ^ p.kin[11]) |
                          (l3.z ^ p.kin[12]) | (l3.w ^ p.kin[13]);
    return p.can && (w0 & ~kW0One) == p.w0want && diff == 0u;
}
// two independent slot lines in flight at once
__device__ __forceinline__ void slot_load2(const uint4 *p, const uint4 *q, uint4 &a, uint4 &b, uint4 &c, uint4 &d,
                                           uint4 &e, uint4 &f, uint4 &g, uint4 &h) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                 "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %9, off sc1\n\t"
                 "global_load_dwordx4 %5, %9, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %6, %9, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %7, %9, off offset:48 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h) : "v"(p), "v"(q) : "memory");
}

// A new key's slot line written as ONE 64-byte store: the lanes that have just won a slot put their line (w0 and the 14 key
// dwords) into a small LDS stage, and the wave writes 16 lines per store instruction, four lanes to a line, 16 bytes each -
// a line is then one whole-line write at the memory side instead of four partial ones (the per-lane form: four
// global_store_dwordx4 sc1 of one lane = four 16-byte writes; an insert was a line read, a CAS and four writes).  Wave-uniform
// call; `won` lanes carry sp / w0 / kin.  stage: kLineStage lines of 16 dwords + their slot pointers, per wave.
#ifndef VF_COOP_STORE
#define VF_COOP_STORE 1
#endif
constexpr int kLineStage = 16;
struct LineStage { u32x4 q[kLineStage][4]; unsigned long long ptr[kLineStage]; };
__device__ __forceinline__ void coop_line_store(LineStage &st, bool won, uint4 *sp, uint64_t w0, const uint32_t (&kin)[kInlineDwords]) {
    uint64_t m = __ballot(won);
    const uint32_t lane = lane_id();
    while (m) {                                                   // (wave-uniform)
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const bool mine = won && ((m >> lane) & 1ull) && rank < (uint32_t)kLineStage;
        if (mine) {
            st.q[rank][0] = u32x4{(uint32_t)w0, (uint32_t)(w0 >> 32), kin[0], kin[1]};
            st.q[rank][1] = u32x4{kin[2], kin[3], kin[4], kin[5]};
            st.q[rank][2] = u32x4{kin[6], kin[7], kin[8], kin[9]};
            st.q[rank][3] = u32x4{kin[10], kin[11], kin[12], kin[13]};
            st.ptr[rank] = (unsigned long long)sp;
        }
        const uint32_t n = min((uint32_t)__popcll(m), (uint32_t)kLineStage);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t w = lane >> 2, q = lane & 3u;
        if (w < n) {
            const u32x4 v = *VF_LDS(u32x4, &st.q[w][q]);
            uint4 *dst = reinterpret_cast<uint4 *>(*VF_LDS(unsigned long long, &st.ptr[w])) + q;
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
        }
        __builtin_amdgcn_wave_barrier();
        // the first n set bits of m are done
        uint64_t done = 0;
        {
            const bool taken = ((m >> lane) & 1ull) && rank < (uint32_t)kLineStage;
            done = __ballot(taken);
        }
        m &= ~done;
    }
}

// U reads per lane and trip (their slot lines are fetched together).  Rows and spans of the next trip are
// prefetched before the current keys go to the table.  Reads whose key was not found by the one-shot
// lookup are queued per wave in LDS and handled 64 at a time by table_find_or_insert, so the insert
// protocol's extra round trips are paid by full waves instead of by every wave that holds one new key.
constexpr int kSlowCap = 192;
#ifndef VF_FINISH_EARLY_PREFETCH
#define VF_FINISH_EARLY_PREFETCH 1      // 0: the next row is fetched only once this trip's key is built (k_finish alone 0.412 against 0.393 ms)
#endif
#ifndef VF_FAST_INSERT_FROM
#define VF_FAST_INSERT_FROM 2
#endif
constexpr uint64_t kFastInsertFromTrip = VF_FAST_INSERT_FROM;   // first trip of a launch on which the one-shot lookup may insert
#ifndef VF_KEYCACHE
#define VF_KEYCACHE 1
#endif
// (register caps through amdgpu_waves_per_eu were tried: 96 registers -> 32 bytes of scratch, two of its reloads on the path of
// every trip: k_finish +15 %, k_scan beside it -8 % with its third wave per SIMD, the step +2 %; 80 registers -> +20 %)
#ifndef VF_FINISH_WAVES
#define VF_FINISH_ATTR __launch_bounds__(kBlock)
#else
#define VF_FINISH_ATTR __launch_bounds__(kBlock, VF_FINISH_WAVES)      // A/B: register cap through the minimum waves per SIMD
#endif
#ifdef VF_FINISH_ABLATION
// diagnosis build: per wave of k_finish, 10 ns ticks (s_memrealtime): [0] start, [1..8] after trip 0,1,2,4,8,16,32,last,
// [9] end, [10] insert-queue batches run, [11] ticks spent in them
__device__ uint64_t g_fin_trace[kMaxGrid * (kBlock / WAVE)][12];
#define VF_FIN_STAMP(i) do { if (lane_id() == 0) g_fin_trace[blockIdx.x * (kBlock / WAVE) + threadIdx.x / WAVE][i] = wall_clock64(); } while (0)
#else
#define VF_FIN_STAMP(i) do { } while (0)
#endif
template <int W, int U, bool DNA, int DBG = 0>
__global__ VF_FINISH_ATTR void k_finish(const uint32_t *__restrict__ rows, const uint16_t *__restrict__ lens,
                                                   uint64_t n_reads, const short2 *__restrict__ spans, Table t, CountLog log) {
    using FK = FinishKey<W, DNA>;
    constexpr int ND = FK::ND;
    __shared__ uint8_t lut[64];
    __shared__ Agg agg;
    __shared__ KeyCache kc;
    __shared__ BlockTable bt;
    __shared__ uint32_t s_stat[4];                // accepted, counted, slow, (slot, count) pairs logged
    __shared__ uint32_t s_slow[kBlock / WAVE][kSlowCap];
    // (one wave-private stage for both: a lookup's image is read back before the trip's new lines are staged, and those have left
    // before the next trip's lookup)
#if VF_COOP_LOOKUP
    union WaveStage { LookupStage lookup; LineStage line; };
#else
    union WaveStage { LineStage line; };
#endif
    __shared__ __attribute__((aligned(16))) unsigned char s_wave_stage[kBlock / WAVE][sizeof(WaveStage)];
#if VF_COOP_LOOKUP
    LookupStage &my_lookup = *reinterpret_cast<LookupStage *>(s_wave_stage[threadIdx.x / WAVE]);
#endif
    LineStage &my_line = *reinterpret_cast<LineStage *>(s_wave_stage[threadIdx.x / WAVE]);
    (void)my_line;
    extern __shared__ uint32_t s_tot[];           // [2][kLogMaxParts] log entries per slot part (log.parts != 0 only)
#ifdef VF_FINISH_PRIO            // A/B: the table kernel's waves issue ahead of the chain kernels' waves on a shared SIMD
    __builtin_amdgcn_s_setprio(VF_FINISH_PRIO);
#endif
    if (threadIdx.x < 64) lut[threadIdx.x] = c_codon_le[threadIdx.x];
    if (threadIdx.x < 4) s_stat[threadIdx.x] = 0;
    if (log.parts) for (uint32_t i = threadIdx.x; i < 2 * kLogMaxParts; i += kBlock) s_tot[i] = 0;
    agg_init(agg);
    keycache_init(kc);
    block_table_begin(bt, t);
    __syncthreads();
    const uint32_t lane = lane_id();
    const uint32_t gwave = blockIdx.x * (kBlock / WAVE) + threadIdx.x / WAVE;
    WaveLog wl{log.one + (uint64_t)gwave * log.cap_one, 0u};
    uint32_t *slow = s_slow[threadIdx.x / WAVE];
    uint32_t nslow = 0;                           // wave-uniform
    const uint64_t gstride = (uint64_t)gridDim.x * kBlock;
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;           // (32 bits per lane; read indices are formed against the wave-uniform trip base)
    const uint64_t n_trips = (n_reads + gstride * U - 1) / (gstride * U);

    short2 sp_next[U];
    uint32_t x_next[U][W + 2];
    auto prefetch = [&](int u, uint64_t rr) {
        if (rr < n_reads) {
            sp_next[u] = spans[rr];
            if constexpr (DBG == 2) {
#pragma unroll
                for (int i = 0; i < W + 2; ++i) x_next[u][i] = (uint32_t)rr * 2654435761u + i;
            } else load_row<W>(rows, rr, x_next[u]);
        }
        else {
            sp_next[u] = make_short2(kNone, kNone);
#pragma unroll
            for (int i = 0; i < W + 2; ++i) x_next[u][i] = 0;
        }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) prefetch(u, (uint64_t)u * gstride + tid);

    VF_FIN_STAMP(0);
#ifdef VF_FINISH_ABLATION
    uint64_t dbg_batches = 0, dbg_ticks = 0;
#endif
#ifdef VF_FINISH_PAD            // A/B only: VF_FINISH_PAD extra registers kept live through the trip loop (how much does the kernel's
    uint32_t pad[VF_FINISH_PAD];   // register footprint cost the kernels that run beside it?)
#pragma unroll
    for (int i = 0; i < VF_FINISH_PAD; ++i) pad[i] = lane * (uint32_t)(i + 3);
#endif
    for (uint64_t trip = 0; trip <= n_trips; ++trip) {
#ifdef VF_FINISH_PAD
#pragma unroll
        for (int i = 0; i < VF_FINISH_PAD; ++i) asm volatile("" : "+v"(pad[i]));
#endif
        const bool last = trip == n_trips;        // the extra trip only drains the slow list
        if (!last) {
            uint32_t x[U][W + 2];
            int start[U], end[U];
            bool ok[U];
            uint32_t r[U];                        // (n_reads < 2^32: launch_finish)
            bool any_ok = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                r[u] = (uint32_t)((trip * U + (uint64_t)u) * gstride + tid);
                start[u] = sp_next[u].x; end[u] = sp_next[u].y;
                ok[u] = start[u] >= 0 && end[u] >= 0 && start[u] < end[u];      // lib.rs:288
#pragma unroll
                for (int i = 0; i < W + 2; ++i) x[u][i] = x_next[u][i];
                any_ok = any_ok || ok[u];
            }
            // The next trip's row is fetched first thing (VF_FINISH_EARLY_PREFETCH=0: only once this trip's key has been built, so
            // that the kernel holds one row at a time - no fewer registers in the end, and 5 % slower).
#if VF_FINISH_EARLY_PREFETCH
#pragma unroll
            for (int u = 0; u < U; ++u) prefetch(u, ((trip + 1) * U + (uint64_t)u) * gstride + tid);
#endif
            if (!__any(any_ok)) {
#if !VF_FINISH_EARLY_PREFETCH
#pragma unroll
                for (int u = 0; u < U; ++u) prefetch(u, ((trip + 1) * U + (uint64_t)u) * gstride + tid);
#endif
            } else {
                uint32_t key[U][ND];
                uint32_t len[U];
                bool act[U];
                FastProbe fp[U];
                uint32_t n_ok = 0, n_act = 0;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    finish_key<W, DNA>(x[u], start[u], end[u], ok[u], lut, key[u], len[u], act[u]);
                    fast_prepare<ND>(t, key[u], FK::kKind, len[u], act[u], fp[u]);
                    n_ok += (uint32_t)__popcll(__ballot(ok[u]));
                    n_act += (uint32_t)__popcll(__ballot(act[u]));
                }
#if !VF_FINISH_EARLY_PREFETCH
#pragma unroll
                for (int u = 0; u < U; ++u) prefetch(u, ((trip + 1) * U + (uint64_t)u) * gstride + tid);
#endif
                if (lane == 0) { atomicAdd(&s_stat[0], n_ok); atomicAdd(&s_stat[1], n_act); }
                bool found[U], cached[U], inserted[U], owner[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    inserted[u] = false;
                    const int role = VF_KEYCACHE ? keycache_probe(kc, fp[u].h1, fp[u].h2, fp[u].meta, fp[u].kin, fp[u].can && fp[u].nd <= 8u) : (int)kCacheNone;
                    cached[u] = role == kCacheCounted;
                    owner[u] = role == kCacheOwner;
                    fp[u].can = fp[u].can && !cached[u];
#ifdef VF_INSERT_COUNTERS
                    { const uint64_t mh = __ballot(cached[u]); if (mh && lane == 0) atomicAdd((unsigned long long *)&t.counters[C_DBG4], (unsigned long long)__popcll(mh)); }
#endif
                }
                if constexpr (U == 2) {
                    uint4 a, b, c, d, e, f, g, h;
                    if constexpr (DBG == 1 || DBG == 3) { found[0] = fp[0].can; found[1] = fp[1].can; }
                    else {
                    slot_load2(t.slots + (uint64_t)fp[0].slot * 4, t.slots + (uint64_t)fp[1].slot * 4, a, b, c, d, e, f, g, h);
                    found[0] = fast_check(fp[0], a, b, c, d);
                    found[1] = fast_check(fp[1], e, f, g, h);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        uint4 a, b, c, d;
                        uint4 *sp = t.slots + (uint64_t)fp[u].slot * 4;
#if VF_COOP_LOOKUP
                        slot_load_coop(my_lookup, t.slots, fp[u].slot, fp[u].nd, fp[u].can, FK::kKind == kKindBytes ? t.key_tag : 0u, a, b, c, d);
#else
                        slot_load_key(sp, fp[u].nd, fp[u].can, FK::kKind == kKindBytes ? t.key_tag : 0u, a, b, c, d);
#endif
                        found[u] = fast_check(fp[u], a, b, c, d);
                        // A new key whose home slot is empty is inserted right here (letter keys: they validate themselves, so
                        // the claim is READY at once and the line is stored without a wait - table_find_or_insert's selfval
                        // case): one CAS round trip for the wave instead of a place in the insert queue, a second load of the
                        // row, a second translation and the claim loop's trips (0.65 M of the 0.8 M new keys of a cold cfg-2
                        // step).  Not on the first trips of a launch: there every lane that holds a hot key still finds the
                        // slot empty, and thousands of CAS on one address serialise (the queue spreads them out).
                        if constexpr (!DNA) {
                            const uint64_t w0seen = (uint64_t)a.x | ((uint64_t)a.y << 32);      // (0, or a word of an earlier epoch)
                            const bool empty = fp[u].can && w0_free(t, w0seen) && trip >= kFastInsertFromTrip;
                            if (__any(empty)) {
                                const uint64_t w0claim = fp[u].w0want | kW0One;
                                unsigned long long old = ~0ull;
                                if (empty) old = atomicCAS(reinterpret_cast<unsigned long long *>(sp), (unsigned long long)w0seen, (unsigned long long)w0claim);
                                const bool won = empty && old == w0seen;
#if VF_COOP_STORE
                                coop_line_store(my_line, won, sp, w0claim, fp[u].kin);
#else
                                if (won) {
                                    const uint32_t (&kin)[kInlineDwords] = fp[u].kin;
                                    slot_store_nowait(sp, u32x4{(uint32_t)w0claim, (uint32_t)(w0claim >> 32), kin[0], kin[1]}, u32x4{kin[2], kin[3], kin[4], kin[5]},
                                                      u32x4{kin[6], kin[7], kin[8], kin[9]}, u32x4{kin[10], kin[11], kin[12], kin[13]});
                                }
#endif
                                if (won) {
                                    atomicAdd(&bt.new_entries, 1u);
                                    atomicAdd(&bt.new_key_dwords, fp[u].nd);
                                }
                                inserted[u] = won;
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    bool to_log = false;
                    if (found[u] && DBG != 3 && DBG != 4 && VF_BOUND(t, fp[u].slot <= t.slot_mask, 4, fp[u].slot, 0)) to_log = !agg_try(agg, fp[u].slot);
                    if (VF_KEYCACHE && owner[u] && (found[u] || inserted[u])) keycache_publish(kc, fp[u].h1, fp[u].h2, fp[u].meta, fp[u].kin, fp[u].slot, true);
                    log_one(wl, log, s_tot, fp[u].slot, to_log);
                    const bool miss = act[u] && !found[u] && !cached[u] && !inserted[u];
                    const uint64_t m = __ballot(miss);
                    if (miss && VF_BOUND(t, nslow + (uint32_t)WAVE <= (uint32_t)kSlowCap, 6, nslow, trip)) slow[nslow + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)r[u];
                    nslow += (uint32_t)__popcll(m);
                    if (m && lane == 0) atomicAdd(&s_stat[2], (uint32_t)__popcll(m));
                    // capacity: < 64 queued before a trip, at most 64 more per read of the trip
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef VF_FINISH_ABLATION
        if (!last) {
            if (trip == 0) VF_FIN_STAMP(1); else if (trip == 1) VF_FIN_STAMP(2); else if (trip == 2) VF_FIN_STAMP(3); else if (trip == 4) VF_FIN_STAMP(4);
            else if (trip == 8) VF_FIN_STAMP(5); else if (trip == 16) VF_FIN_STAMP(6); else if (trip == 32) VF_FIN_STAMP(7);
            if (trip + 1 == n_trips) VF_FIN_STAMP(8);
        }
        const uint64_t dbg_t0 = wall_clock64();
        const uint32_t dbg_n0 = nslow;
#endif
        // The insert queue's claim loop is where the kernel's registers peak (a slot line, the key, the row it rebuilds the key
        // from).  The prefetched row of the next trip is given up for its duration and fetched again behind it - CFG-explicit
        // (if + do-while + reload), so that the row is not live across the loop, and the LONG-key paths are compiled out for keys
        // that cannot be long (RegKey::kMaxDwords).  The kernel stands at 125 registers (103 without the key cache, 108 without
        // this loop); three of its waves leave a SIMD room for one k_scan wave of the other batch lane, and that is not what keeps
        // the chain kernels slow beside it: padded to 157 registers - no room at all - k_scan slows by 3 %, capped at 96 it
        // gains 5 % and this kernel loses 30 % (EXPERIMENTS R3.6).  A wave runs the loop on ~3 of its 51 trips.
        if (nslow >= (last ? 1u : (uint32_t)WAVE)) {
        do {
#ifdef VF_FINISH_ABLATION
            ++dbg_batches;
#endif
            const uint32_t cnt = nslow < (uint32_t)WAVE ? nslow : (uint32_t)WAVE;
            const uint32_t base = nslow - cnt;
            const bool mine = lane < cnt;
            const uint64_t rr = mine ? (uint64_t)VF_LDS(uint32_t, slow)[base + lane] : 0ull;
            uint32_t xs[W + 2];
            short2 sp = make_short2(kNone, kNone);
            if (mine && VF_BOUND(t, rr < n_reads, 7, rr, base)) { sp = spans[rr]; load_row<W>(rows, rr, xs); }
            else {
#pragma unroll
                for (int i = 0; i < W + 2; ++i) xs[i] = 0;
            }
            const bool oks = mine && sp.x >= 0 && sp.y >= 0 && sp.x < sp.y;
            uint32_t ks[ND];
            uint32_t ls;
            bool as;
            finish_key<W, DNA>(xs, sp.x, sp.y, oks, lut, ks, ls, as);
            bool counted;
#ifdef VF_DIAG_NO_SLOW          // register diagnosis only (wrong results): where is the kernel's register peak?
            const uint32_t sl = ks[0] ^ ls; counted = as;
#else
            const uint32_t sl = table_find_or_insert(t, bt, RegKey<ND>{ks}, FK::kKind, ls, as, kNoOff, true, counted);
#endif
            if (VF_KEYCACHE) {                  // the key's cache entry may be waiting for this slot
                FastProbe fq;
                fast_prepare<ND>(t, ks, FK::kKind, ls, as, fq);
                keycache_publish(kc, fq.h1, fq.h2, fq.meta, fq.kin, sl, fq.can && fq.nd <= 8u);
            }
            log_one(wl, log, s_tot, sl, sl != kNoSlot && VF_BOUND(t, sl <= t.slot_mask, 5, sl, ls) && !counted && !agg_try(agg, sl));
            nslow = base;
            __builtin_amdgcn_wave_barrier();
        } while (nslow >= (last ? 1u : (uint32_t)WAVE));
#pragma unroll
        for (int u = 0; u < U; ++u) prefetch(u, last ? n_reads : ((trip + 1) * U + (uint64_t)u) * gstride + tid);
        }
#ifdef VF_FINISH_ABLATION
        if (dbg_n0 != nslow) dbg_ticks += wall_clock64() - dbg_t0;
#endif
    }
#ifdef VF_FINISH_ABLATION
    VF_FIN_STAMP(9);
    if (lane == 0) { g_fin_trace[gwave][10] = dbg_batches; g_fin_trace[gwave][11] = dbg_ticks; }
#endif
#ifdef VF_FINISH_PAD
    {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < VF_FINISH_PAD; ++i) acc ^= pad[i];
        if (acc == 0xFFFFFFFFu) slow[0] = acc;           // (never true: keeps the pad alive)
    }
#endif
    if (lane == 0) log.n_one[gwave] = wl.n < log.cap_one ? wl.n : log.cap_one;
    __syncthreads();
    agg_flush(agg, log, s_tot, &s_stat[3]);
    keycache_flush(kc, t, log, s_tot, &s_stat[3]);
    block_table_end(bt, t);
    __syncthreads();
    if (log.parts)
        for (uint32_t i = threadIdx.x; i < 2 * kLogMaxParts; i += kBlock) {
            const uint32_t v = s_tot[i];
            if (v) atomicAdd(&log.tot[i], v);
        }
    if (threadIdx.x == 0) {
        log.n_many[blockIdx.x] = s_stat[3] < kLogMany ? s_stat[3] : kLogMany;
        if (s_stat[0]) atomicAdd((unsigned long long *)&t.counters[C_ACCEPTED], (unsigned long long)s_stat[0]);
        if (s_stat[1]) atomicAdd((unsigned long long *)&t.counters[C_COUNTED], (unsigned long long)s_stat[1]);
        if (s_stat[2]) atomicAdd((unsigned long long *)&t.counters[C_SLOW], (unsigned long long)s_stat[2]);
    }
}

// ------------------------------------------------------------------------------------------------
// the count log applied (CountLog, vf_device.h): the reduce of lib.rs:292-306 for a whole batch
// ------------------------------------------------------------------------------------------------
// exclusive prefix of v[0..n) over the workgroup (n <= kLogMaxParts <= kBlock), result in out[0..n], out[n] = total
__device__ __forceinline__ void part_starts(const uint32_t *__restrict__ tot, uint32_t n, uint32_t *out) {
    static_assert(kLogMaxParts <= (uint32_t)kBlock, "one thread per part");
    __shared__ uint32_t s_wave[kBlock / WAVE];
    const uint32_t i = threadIdx.x;
    const uint32_t v = i < n ? tot[i] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if ((int)lane_id() >= d) incl += up;
    }
    if (lane_id() == WAVE - 1) s_wave[i / WAVE] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < i / WAVE; ++w) base += s_wave[w];
    if (i < n) out[i] = base + incl - v;
    if (i == kBlock - 1) out[n] = base + incl;
    __syncthreads();
}

// One workgroup per workgroup of k_finish: its four wave segments and its pairs, grouped by slot part.  A run per part is
// reserved in the part's stretch of the output with one atomic (the totals say where the stretch starts).  The single
// occurrences - the bulk: ~3 M a cfg-2 step - are first put in part order in LDS and leave as whole runs, consecutive lanes to
// consecutive addresses: placed one by one (rounds 3) every entry was a 4-byte write request of its own at the L2, 2.7 M of
// them a step (profiles/r04_requests.json), and the step follows the number of L2 requests (EXPERIMENTS R4.4).
constexpr uint32_t kScatterChunk = 4096;                  // entries sorted in LDS at a time
__global__ __launch_bounds__(kBlock) void k_log_scatter(CountLog log, uint64_t *__restrict__ counters) {
    __shared__ uint32_t s_start[2][kLogMaxParts + 1];     // part starts (one, many)
    __shared__ uint32_t s_n[2][kLogMaxParts];             // this workgroup's entries per part, then its cursor
    __shared__ uint32_t s_base[2][kLogMaxParts];
    __shared__ uint32_t s_loc[kLogMaxParts + 1];          // the chunk's part starts inside s_sorted
    __shared__ uint32_t s_sorted[kScatterChunk];
    const uint32_t P = log.parts;
    part_starts(log.tot, P, s_start[0]);
    part_starts(log.tot + kLogMaxParts, P, s_start[1]);
    const uint32_t b = blockIdx.x;
    bool bad = false;
    // ---- single occurrences: the four wave segments as one sequence, a chunk at a time
    uint32_t seg_n[kBlock / WAVE], total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / WAVE; ++w) { seg_n[w] = log.n_one[b * (kBlock / WAVE) + w]; total += seg_n[w]; }
    auto entry = [&](uint32_t i) -> uint32_t {            // i-th entry of the workgroup (i < total)
        uint32_t w = 0;
#pragma unroll
        for (uint32_t k = 0; k + 1 < kBlock / WAVE; ++k) if (w == k && i >= seg_n[k]) { i -= seg_n[k]; w = k + 1; }
        return log.one[(uint64_t)(b * (kBlock / WAVE) + w) * log.cap_one + i];
    };
    for (uint32_t c0 = 0; c0 < total; c0 += kScatterChunk) {              // (workgroup-uniform)
        const uint32_t cn = min(kScatterChunk, total - c0);
        for (uint32_t i = threadIdx.x; i < kLogMaxParts; i += kBlock) s_n[0][i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cn; i += kBlock) atomicAdd(&s_n[0][entry(c0 + i) >> log.shift], 1u);
        __syncthreads();
        part_starts(s_n[0], P, s_loc);                                     // (ends with a barrier)
        for (uint32_t p = threadIdx.x; p < P; p += kBlock) {
            const uint32_t n = s_n[0][p];
            s_base[0][p] = s_start[0][p] + (n ? atomicAdd(&log.cur[p], n) : 0u);
            s_n[0][p] = 0;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cn; i += kBlock) {
            const uint32_t slot = entry(c0 + i), p = slot >> log.shift;
            s_sorted[s_loc[p] + atomicAdd(&s_n[0][p], 1u)] = slot;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cn; i += kBlock) {
            const uint32_t slot = s_sorted[i], p = slot >> log.shift;
            const uint64_t at = (uint64_t)s_base[0][p] + (i - s_loc[p]);
            if (at < log.cap_total) log.one_by_part[at] = slot; else bad = true;
        }
        __syncthreads();
    }
    // ---- (slot, count) pairs: a few hundred a workgroup, placed one by one
    for (uint32_t i = threadIdx.x; i < kLogMaxParts; i += kBlock) s_n[1][i] = 0;
    __syncthreads();
    const uint32_t n_many = log.n_many[b];
    const uint2 *many = log.many + (uint64_t)b * kLogMany;
    for (uint32_t i = threadIdx.x; i < n_many; i += kBlock) atomicAdd(&s_n[1][many[i].x >> log.shift], 1u);
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < P; p += kBlock) {
        const uint32_t n = s_n[1][p];
        s_base[1][p] = s_start[1][p] + (n ? atomicAdd(&log.cur[kLogMaxParts + p], n) : 0u);
        s_n[1][p] = 0;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_many; i += kBlock) {
        const uint2 e = many[i];
        const uint32_t p = e.x >> log.shift;
        const uint64_t at = (uint64_t)s_base[1][p] + atomicAdd(&s_n[1][p], 1u);
        if (at < (uint64_t)log.grid * kLogMany) log.many_by_part[at] = e; else bad = true;
    }
    if (bad) atomicOr((unsigned long long *)&counters[C_ERROR], 8ull);      // totals and log disagree: never seen, never silent
}

// One workgroup per range of kLogRange slots: the entries of the range's part are counted in LDS (the part's other
// ranges are skipped over: a part is read by every workgroup of its ranges, from L2), then the counts go to the table's
// 64-bit words, consecutive addresses, once per slot and batch.
__global__ __launch_bounds__(kBlock) void k_log_apply(Table t, CountLog log) {
    __shared__ uint32_t s_h[kLogRange];
    __shared__ uint32_t s_start[2][kLogMaxParts + 1];
    const uint32_t P = log.parts;
    part_starts(log.tot, P, s_start[0]);
    part_starts(log.tot + kLogMaxParts, P, s_start[1]);
    const uint64_t nslots = (uint64_t)t.slot_mask + 1;
    const uint32_t R = nslots < kLogRange ? (uint32_t)nslots : kLogRange;
    const uint64_t n_ranges = nslots / R;
    for (uint64_t r = blockIdx.x; r < n_ranges; r += gridDim.x) {
        const uint32_t lo = (uint32_t)(r * R), p = lo >> log.shift;
        const uint32_t a1 = s_start[0][p], b1 = s_start[0][p + 1], a2 = s_start[1][p], b2 = s_start[1][p + 1];
        if (a1 == b1 && a2 == b2) continue;                           // (workgroup-uniform)
        for (uint32_t i = threadIdx.x; i < R; i += kBlock) s_h[i] = 0;
        __syncthreads();
        for (uint32_t i = a1 + threadIdx.x; i < b1; i += kBlock) {
            const uint32_t d = log.one_by_part[i] - lo;
            if (d < R) atomicAdd(&s_h[d], 1u);
        }
        for (uint32_t i = a2 + threadIdx.x; i < b2; i += kBlock) {
            const uint2 e = log.many_by_part[i];
            const uint32_t d = e.x - lo;
            if (d < R) atomicAdd(&s_h[d], e.y);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < R; i += kBlock) {
            const uint32_t h = s_h[i];
            if (h) base_add(t, lo + i, (unsigned long long)h);
        }
        __syncthreads();
    }
}

// parts == 0 (a table of so many ranges that a batch touches few of them): entry by entry
__global__ __launch_bounds__(kBlock) void k_log_direct(Table t, CountLog log) {
    const uint32_t b = blockIdx.x;
    for (uint32_t w = 0; w < kBlock / WAVE; ++w) {
        const uint32_t gw = b * (kBlock / WAVE) + w, n = log.n_one[gw];
        const uint32_t *seg = log.one + (uint64_t)gw * log.cap_one;
        for (uint32_t i = threadIdx.x; i < n; i += kBlock) base_add(t, seg[i], 1ull);
    }
    const uint32_t n_many = log.n_many[b];
    const uint2 *many = log.many + (uint64_t)b * kLogMany;
    for (uint32_t i = threadIdx.x; i < n_many; i += kBlock) base_add(t, many[i].x, (unsigned long long)many[i].y);
}

// ------------------------------------------------------------------------------------------------
// k_generic: byte-wise restatement of the whole per-read function (any byte, any length)
// workspace ws[row*stride + thread]: rows 0..4m-1 = H,HL,E,EL; rows 4m.. = key staging
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int g_code(uint8_t c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}
__device__ __forceinline__ int g_tr_index(uint8_t c) {
    if (c == 'U' || c == 'u') return 3;
    return g_code(c);
}
__device__ int64_t g_find_exact(const uint8_t *s, int64_t n, const uint8_t *a, int64_t m) {
    if (m == 0) return 0;
    for (int64_t k = 0; k + m <= n; ++k) {
        int64_t j = 0;
        while (j < m && s[k + j] == a[j]) ++j;
        if (j == m) return k;
    }
    return -1;
}
// returns false when no cell exists (n == 0)
__device__ bool g_sg(const uint8_t *q, int m, const uint8_t *r, int64_t n, const GenericParams &gp, int32_t *ws,
                     uint64_t st, int64_t &score, int64_t &len) {
    if (n <= 0 || m <= 0) return false;
    const int32_t NEG = INT32_MIN / 4;
    int32_t *H = ws, *HL = ws + (uint64_t)m * st, *E = ws + 2ull * m * st, *EL = ws + 3ull * m * st;
    for (int i = 0; i < m; ++i) { H[i * st] = 0; HL[i * st] = 0; E[i * st] = NEG; EL[i * st] = 0; }
    int32_t best = NEG, bestl = 0;
    for (int64_t j = 0; j < n; ++j) {
        const int rc = g_code(r[j]);
        int32_t hdiag = 0, hldiag = 0, hup = 0, hlup = 0, f = NEG, fl = 0;
        for (int i = 0; i < m; ++i) {
            const int32_t hc = H[i * st], hlc = HL[i * st];
            const int32_t eo = hc - gp.open, ee = E[i * st] - gp.ext;
            int32_t e, el;
            if (eo > ee) { e = eo; el = hlc + 1; } else { e = ee; el = EL[i * st] + 1; }
            const int32_t fo = hup - gp.open, fe = f - gp.ext;
            if (fo > fe) { f = fo; fl = hlup + 1; } else { f = fe; fl = fl + 1; }
            const int qc = g_code(q[i]);
            const int32_t sub = (qc == 4 || rc == 4) ? 0 : (qc == rc ? gp.match : gp.mismatch);
            const int32_t d = hdiag + sub, dl = hldiag + 1;
            int32_t h = max(d, max(e, f));
            const int32_t hl = (h == d) ? dl : ((h == f) ? fl : el);
            hdiag = hc; hldiag = hlc;
            H[i * st] = h; HL[i * st] = hl; E[i * st] = e; EL[i * st] = el;
            hup = h; hlup = hl;
        }
        if (hup > best) { best = hup; bestl = hlup; }
    }
    for (int i = 0; i < m - 1; ++i)
        if (H[i * st] > best) { best = H[i * st]; bestl = HL[i * st]; }
    score = best; len = bestl;
    return true;
}
__device__ int64_t g_adapter_match(const uint8_t *s, int64_t n, const uint8_t *a, int m, uint32_t align, int64_t T,
                                   const GenericParams &gp, int32_t *ws, uint64_t st, bool is_prefix) {
    const int64_t k = g_find_exact(s, n, a, m);
    if (k >= 0) return is_prefix ? k + m : k;
    if (!align) return -1;
    int64_t score, len;
    if (!g_sg(a, m, s, n, gp, ws, st, score, len)) return -1;
    if (score >= T) {
        if (is_prefix) return len;
        return len <= n ? n - len : -1;
    }
    return -1;
}

__global__ __launch_bounds__(kBlock) void k_generic(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ offs,
                                                    uint64_t n_reads, GenericParams gp, int32_t *__restrict__ starts,
                                                    int32_t *__restrict__ ends, int32_t *__restrict__ ws, uint64_t st,
                                                    uint32_t ws_rows, Table t, const uint32_t *__restrict__ idx_start,
                                                    const uint32_t *__restrict__ idx_len, const uint32_t *__restrict__ n_dev) {
    // reads either lie back to back (offs[r] .. offs[r+1]) or are listed by the device framer (idx_start/idx_len,
    // their number in *n_dev)
    if (n_dev) n_reads = *n_dev;
    __shared__ uint32_t s_stat[2];
    __shared__ BlockTable bt;
    if (threadIdx.x < 2) s_stat[threadIdx.x] = 0;
    block_table_begin(bt, t);
    __syncthreads();
    const uint64_t tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;   // < st by construction
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t n_round = (n_reads + WAVE - 1) / WAVE * WAVE;
    int32_t *my = ws + tid;
    const uint32_t mmax = max(gp.pre_len, gp.suf_len);
    uint32_t *stage = reinterpret_cast<uint32_t *>(my + 4ull * mmax * st);
    for (uint64_t r = tid; r < n_round; r += stride) {
        const bool valid = r < n_reads;
        const uint8_t *s = bytes;
        int64_t n = 0, a = -1, b = -1;
        if (valid) {
            if (idx_start) { s = bytes + idx_start[r]; n = (int64_t)idx_len[r]; }
            else { s = bytes + offs[r]; n = (int64_t)(offs[r + 1] - offs[r]); }
            a = g_adapter_match(s, n, gp.pre, (int)gp.pre_len, gp.align_pre, gp.T_pre, gp, my, st, true);
            b = g_adapter_match(s, n, gp.suf, (int)gp.suf_len, gp.align_suf, gp.T_suf, gp, my, st, false);
            if (starts) { starts[r] = (int32_t)a; ends[r] = (int32_t)b; }
        }
        if (!gp.count_mode) continue;
        const bool ok = valid && a >= 0 && b >= 0 && a < b;
        wave_count_lds(&s_stat[0], ok);
        const int64_t L = ok ? b - a : 0;
        const uint8_t *reg = s + (ok ? a : 0);
        uint32_t kind = kKindBytes, klen = 0;
        bool ins = false;
        if (ok && gp.skip_translation) {
            bool pure = true;
            for (int64_t i = 0; i < L; ++i) { const uint8_t c = reg[i]; pure = pure && (c == 'A' || c == 'C' || c == 'G' || c == 'T'); }
            ins = L <= 0xFFFF;      // key length field is 16 bits; longer regions flag an error below
            if (!ins) atomicOr((unsigned long long *)&t.counters[C_ERROR], 4ull);
            if (ins && pure) {
                kind = kKindPacked; klen = (uint32_t)L;
                for (uint32_t d = 0; d < (klen + 15u) >> 4; ++d) {
                    uint32_t v = 0;
                    for (uint32_t q = 0; q < 16 && 16 * d + q < klen; ++q) v |= (uint32_t)g_code(reg[16 * d + q]) << (2 * q);
                    stage[d * st] = v;
                }
            } else if (ins) {
                kind = kKindBytes; klen = (uint32_t)L;
                for (uint32_t d = 0; d < (klen + 3u) >> 2; ++d) {
                    uint32_t v = 0;
                    for (uint32_t q = 0; q < 4 && 4 * d + q < klen; ++q) v |= (uint32_t)reg[4 * d + q] << (8 * q);
                    stage[d * st] = v;
                }
            }
        } else if (ok && (L % 3) == 0) {
            ins = L / 3 <= 0xFFFF;
            if (!ins) atomicOr((unsigned long long *)&t.counters[C_ERROR], 4ull);
            if (ins) {
                kind = kKindBytes; klen = (uint32_t)(L / 3);
                for (uint32_t d = 0; d < (klen + 3u) >> 2; ++d) {
                    uint32_t v = 0;
                    for (uint32_t q = 0; q < 4 && 4 * d + q < klen; ++q) {
                        const uint8_t *c = reg + 3ull * (4 * d + q);
                        uint8_t aa = 'X';
                        if (c[0] < 0x80 && c[1] < 0x80 && c[2] < 0x80) {
                            const int i0 = g_tr_index(c[0]), i1 = g_tr_index(c[1]), i2 = g_tr_index(c[2]);
                            if (i0 != 4 && i1 != 4 && i2 != 4) aa = c_codon_le[i0 + 4 * i1 + 16 * i2];
                        }
                        v |= (uint32_t)aa << (8 * q);
                    }
                    stage[d * st] = v;
                }
            }
        }
        (void)ws_rows;
        wave_count_lds(&s_stat[1], ins);
        table_add(t, bt, MemKey{stage, st}, kind, klen, 1ull, ins);
    }
    __syncthreads();
    block_table_end(bt, t);
    if (threadIdx.x == 0) {
        if (s_stat[0]) atomicAdd((unsigned long long *)&t.counters[C_ACCEPTED], (unsigned long long)s_stat[0]);
        if (s_stat[1]) atomicAdd((unsigned long long *)&t.counters[C_COUNTED], (unsigned long long)s_stat[1]);
    }
}

// ------------------------------------------------------------------------------------------------
// FASTQ text on the device (vf_run_fastq, device framing): newline index, four-line framing, 2-bit packing.
// The host only inflates and copies text; what seq_io's reader and the `work` closure's slicing do
// (lib.rs:234, 271-287) happens here for a whole chunk of records at once.
//   k_nl_count   newlines per 4 KiB tile
//   k_nl_scan    exclusive prefix over the tiles (one workgroup), total line count
//   k_nl_write   position of every newline, in order; longest line
//   k_frame_pack one record per lane: '@' / '+' / equal-length checks (first offending record wins, same
//                order of checks as the host framer), sequence -> 2-bit row; reads the 2-bit path cannot
//                hold (non-ACGT byte, longer than the row) are listed for k_generic and packed as empty
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kTileBytes = 4096;                      // kBlock lanes x 16 bytes
__device__ __forceinline__ uint32_t nl_mask16(const uint8_t *__restrict__ text, uint64_t n, uint64_t at) {
    uint32_t m = 0;
    if (at + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + at);      // text buffers are 16-byte aligned
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 4; ++b) m |= (((w[q] >> (8 * b)) & 0xFFu) == 0x0Au ? 1u : 0u) << (4 * q + b);
    } else {
        for (uint64_t k = at; k < n; ++k) m |= (text[k] == 0x0A ? 1u : 0u) << (uint32_t)(k - at);
    }
    return m;
}
__global__ __launch_bounds__(kBlock) void k_nl_count(const uint8_t *__restrict__ text, uint64_t n, uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t s_c;
    if (threadIdx.x == 0) s_c = 0;
    __syncthreads();
    const uint64_t at = (uint64_t)blockIdx.x * kTileBytes + (uint64_t)threadIdx.x * 16u;
    const uint32_t c = at < n ? (uint32_t)__popc(nl_mask16(text, n, at)) : 0u;
    uint32_t w = c;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) w += __shfl_xor(w, o);
    if (lane_id() == 0 && w) atomicAdd(&s_c, w);
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_c;
}
// info[0] = lines, info[1] = longest line, info[2] = first error ((record << 8) | code, ~0 = none), info[3] = generic reads
__global__ __launch_bounds__(1024) void k_nl_scan(const uint32_t *__restrict__ tile_cnt, uint32_t n_tiles, uint32_t *__restrict__ tile_off,
                                                  uint32_t *__restrict__ info) {
    __shared__ uint32_t s_part[1024];
    const uint32_t per = (n_tiles + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per, hi = min(n_tiles, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += tile_cnt[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? s_part[threadIdx.x - o] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; ++i) { tile_off[i] = run; run += tile_cnt[i]; }
    if (threadIdx.x == 1023) { info[0] = s_part[1023]; info[1] = 0; info[2] = 0xFFFFFFFFu; info[3] = 0; }
}
__global__ __launch_bounds__(kBlock) void k_nl_write(const uint8_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ tile_off,
                                                     uint32_t *__restrict__ line_end) {
    __shared__ uint32_t s_scan[kBlock];
    const uint64_t at = (uint64_t)blockIdx.x * kTileBytes + (uint64_t)threadIdx.x * 16u;
    uint32_t m = at < n ? nl_mask16(text, n, at) : 0u;
    const uint32_t c = (uint32_t)__popc(m);
    s_scan[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t o = 1; o < (uint32_t)kBlock; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? s_scan[threadIdx.x - o] : 0u;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t out = tile_off[blockIdx.x] + s_scan[threadIdx.x] - c;
    while (m) {
        const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
        m &= m - 1u;
        line_end[out++] = (uint32_t)(at + b);
    }
}
__global__ __launch_bounds__(kBlock) void k_line_max(const uint32_t *__restrict__ line_end, const uint32_t *__restrict__ info_in, uint32_t *__restrict__ info) {
    const uint32_t n_lines = info_in[0];
    uint32_t mx = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_lines; i += gridDim.x * kBlock) {
        const uint32_t st = i ? line_end[i - 1] + 1u : 0u;
        mx = max(mx, line_end[i] - st);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    if (lane_id() == 0 && mx) atomicMax(&info[1], mx);
}

enum { kFqBlank = 1, kFqNoAt = 2, kFqNoPlus = 3, kFqLenDiffer = 4 };
template <int W>
__global__ __launch_bounds__(kBlock) void k_frame_pack(const uint8_t *__restrict__ text, const uint32_t *__restrict__ line_end, uint32_t n_rec,
                                                       uint32_t *__restrict__ rows, uint16_t *__restrict__ lens,
                                                       uint32_t *__restrict__ gen_start, uint32_t *__restrict__ gen_len, uint32_t *__restrict__ info,
                                                       int all_generic) {
    for (uint32_t r = blockIdx.x * kBlock + threadIdx.x; r < n_rec; r += gridDim.x * kBlock) {
        const uint32_t l0 = 4u * r;
        const uint32_t s0 = l0 ? line_end[l0 - 1] + 1u : 0u, e0 = line_end[l0];
        const uint32_t s1 = e0 + 1u, e1 = line_end[l0 + 1];
        const uint32_t s2 = e1 + 1u, e2 = line_end[l0 + 2];
        const uint32_t s3 = e2 + 1u, e3 = line_end[l0 + 3];
        uint32_t code = 0;
        if (e0 == s0 || (e0 == s0 + 1u && text[s0] == '\r')) code = kFqBlank;
        else if (text[s0] != '@') code = kFqNoAt;
        uint32_t sl = e1 - s1, ql = e3 - s3;
        if (sl && text[e1 - 1] == '\r') --sl;
        if (ql && text[e3 - 1] == '\r') --ql;
        if (!code && (e2 == s2 || text[s2] != '+')) code = kFqNoPlus;
        if (!code && sl != ql) code = kFqLenDiffer;
        if (code) { atomicMin(&info[2], (r << 8) | code); sl = 0; }
        // 2-bit pack (A0 C1 G2 T3: x ^ (x >> 1) with x = bits 2:1 of the letter); anything else -> k_generic
        uint32_t *row = rows + (uint64_t)r * W;
        bool fast = !all_generic && !code && sl <= 16u * (uint32_t)W;
        if (fast) {
            const uint8_t *p = text + s1;
#pragma unroll 1
            for (uint32_t w = 0; w < (uint32_t)W; ++w) {
                uint32_t v = 0;
                const uint32_t base = 16u * w;
                if (base < sl) {
                    const uint32_t cnt = min(16u, sl - base);
                    for (uint32_t k = 0; k < cnt; ++k) {
                        const uint32_t c = p[base + k];
                        fast = fast && (c == 'A' || c == 'C' || c == 'G' || c == 'T');
                        const uint32_t x = (c >> 1) & 3u;
                        v |= (x ^ (x >> 1)) << (2u * k);
                    }
                }
                row[w] = v;
            }
        }
        if (!fast) {
#pragma unroll 1
            for (uint32_t w = 0; w < (uint32_t)W; ++w) row[w] = 0;
        }
        if (!fast && !code) {
            const uint32_t g = atomicAdd(&info[3], 1u);
            gen_start[g] = s1;
            gen_len[g] = sl;
        }
        lens[r] = fast ? (uint16_t)sl : (uint16_t)0;
    }
}

// ------------------------------------------------------------------------------------------------
// table export (compaction) and merge (multi-GPU exchange)
// ------------------------------------------------------------------------------------------------
// Owner of a key in the exchange between G GPUs (the multi-GPU reduce of lib.rs:292-306): a multiply-shift range
// reduction of the key's 32-bit hash tag (the low word of w0), independent of the slot index and of the table's size.
__device__ __forceinline__ uint32_t key_owner(uint64_t w0, uint32_t G) { return (uint32_t)(((uint64_t)(uint32_t)w0 * G) >> 32); }

// entries and key dwords per owner: out[2 * o], out[2 * o + 1]   (G <= kMaxOwners)
constexpr uint32_t kMaxOwners = 64;
__global__ __launch_bounds__(kBlock) void k_owner_count(Table t, uint32_t G, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long s_h[2 * kMaxOwners];
    for (uint32_t i = threadIdx.x; i < 2 * kMaxOwners; i += kBlock) s_h[i] = 0;
    __syncthreads();
    const uint64_t nslots = (uint64_t)t.slot_mask + 1;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t s = (uint64_t)blockIdx.x * kBlock + threadIdx.x; s < nslots; s += stride) {
        const uint2 h = *reinterpret_cast<const uint2 *>(t.slots + s * 4);
        const uint64_t w0 = (uint64_t)h.x | ((uint64_t)h.y << 32);
        if (w0_live(t, w0)) {
            const uint32_t meta = (uint32_t)(w0 >> 32) & 0x3FFFFu;
            const uint32_t o = key_owner(w0, G);
            atomicAdd(&s_h[2 * o], 1ull);
            atomicAdd(&s_h[2 * o + 1], (unsigned long long)key_dwords(meta >> 16, meta & 0xFFFFu));
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 2 * G; i += kBlock) if (s_h[i]) atomicAdd(&out[i], s_h[i]);
}

// G = 0: every live entry; otherwise only the entries whose key belongs to `owner` of G
__global__ __launch_bounds__(kBlock) void k_export(Table t, uint64_t *__restrict__ out_refs, uint64_t *__restrict__ out_counts,
                                                   uint32_t *__restrict__ out_arena, uint32_t G, uint32_t owner) {
    const uint64_t nslots = (uint64_t)t.slot_mask + 1;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t s0 = (uint64_t)blockIdx.x * kBlock; s0 < nslots; s0 += stride) {
        const uint64_t s = s0 + threadIdx.x;
        uint4 l0 = make_uint4(0, 0, 0, 0), l1 = l0, l2 = l0, l3 = l0;
        bool live = false;
        if (s < nslots) {
            const uint4 *sp = t.slots + s * 4;
            l0 = sp[0];
            const uint64_t w0 = (uint64_t)l0.x | ((uint64_t)l0.y << 32);
            live = w0_live(t, w0) && (G == 0u || key_owner(w0, G) == owner);
            if (live) { l1 = sp[1]; l2 = sp[2]; l3 = sp[3]; }
        }
        const uint64_t m = __ballot(live);
        if (!m) continue;
        const uint64_t w0 = (uint64_t)l0.x | ((uint64_t)l0.y << 32);
        const uint32_t meta = (uint32_t)(w0 >> 32) & 0x3FFFFu;
        const uint32_t nd = live ? key_dwords(meta >> 16, meta & 0xFFFFu) : 0u;
        // wave prefix sums: entry rank and key offset
        uint32_t incl = nd;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if ((int)lane_id() >= d) incl += up;
        }
        const uint32_t total = __shfl(incl, WAVE - 1);
        const uint32_t leader = __ffsll((unsigned long long)m) - 1;
        uint64_t base_e = 0, base_k = 0;
        if (lane_id() == leader) {
            base_e = atomicAdd((unsigned long long *)&t.counters[C_EXPORT_N], (unsigned long long)__popcll(m));
            base_k = atomicAdd((unsigned long long *)&t.counters[C_EXPORT_KEYS], (unsigned long long)total);
        }
        base_e = __shfl(base_e, leader);
        base_k = __shfl(base_k, leader);
        if (live) {
            const uint64_t e = base_e + __popcll(m & ((1ull << lane_id()) - 1ull));
            const uint64_t ko = base_k + (incl - nd);
            out_refs[e] = (ko << 24) | kRefValid | (uint64_t)meta;
            out_counts[e] = count_of(t, s, w0);
            if (w0 & kW0Long) {
                const uint64_t off = ((uint64_t)l0.z | ((uint64_t)l0.w << 32)) & kLongOffMask;
                for (uint32_t d = 0; d < nd; ++d) out_arena[ko + d] = t.arena[off + d];
            } else {
                const uint32_t kk[kInlineDwords] = {l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2.x, l2.y, l2.z, l2.w, l3.x, l3.y, l3.z, l3.w};
                const uint32_t strip = (t.key_tag && (meta >> 16) == kKindBytes) ? ~kTagBits : 0xFFFFFFFFu;      // the epoch pattern is the slot's business
#pragma unroll
                for (int d = 0; d < kInlineDwords; ++d) if ((uint32_t)d < nd) out_arena[ko + d] = kk[d] & strip;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_merge(Table t, const uint64_t *__restrict__ refs, const uint64_t *__restrict__ counts,
                                                  const uint32_t *__restrict__ arena, uint64_t n) {
    __shared__ BlockTable bt;
    block_table_begin(bt, t);
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t n_round = (n + WAVE - 1) / WAVE * WAVE;
    for (uint64_t e = (uint64_t)blockIdx.x * kBlock + threadIdx.x; e < n_round; e += stride) {
        const bool valid = e < n;
        uint64_t ref = 0, cnt = 0;
        if (valid) { ref = refs[e]; cnt = counts[e]; }
        const uint32_t len = (uint32_t)(ref & 0xFFFFull), kind = (uint32_t)((ref >> 16) & 3ull);
        // A table cleared by epoch stores letter keys with the epoch in their bytes' top bits and strips those bits on the way
        // out: a byte-wise key that uses them (a multi-byte UTF-8 region out of a skip_translation context's export, a malformed
        // peer blob) would come back corrupted, and a reader of its line would wait for ever for "written" dwords.  Such a key
        // is refused, loudly (C_ERROR bit 5 -> VF_EINVAL), never stored.
        bool storable = valid;
        if (valid && t.key_tag != 0u && kind == kKindBytes) {
            uint32_t high = 0;
            const uint32_t nd = key_dwords(kind, len);
            for (uint32_t d = 0; d < nd; ++d) high |= arena[(ref >> 24) + d] & kTagBits;
            if (high) { storable = false; atomicOr((unsigned long long *)&t.counters[C_ERROR], 32ull); }
        }
        table_add(t, bt, MemKey{arena + (ref >> 24), 1ull}, kind, len, cnt, storable);
    }
    __syncthreads();
    block_table_end(bt, t);
}

// every entry of `from` re-inserted into `to` (table growth); both share arena, counters and chunks; arena keys stay in place
__global__ __launch_bounds__(kBlock) void k_rehash(Table from, Table to) {
    __shared__ BlockTable bt;
    block_table_begin(bt, to);
    __syncthreads();
    const uint64_t nslots = (uint64_t)from.slot_mask + 1;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t s0 = (uint64_t)blockIdx.x * kBlock; s0 < nslots; s0 += stride) {
        const uint64_t s = s0 + threadIdx.x;
        bool live = false;
        uint64_t w0 = 0, cnt = 0, off = 0;
        const uint32_t *kp = from.arena;
        if (s < nslots) {
            const uint4 *sp = from.slots + s * 4;
            const uint4 l0 = sp[0];
            w0 = (uint64_t)l0.x | ((uint64_t)l0.y << 32);
            live = w0_live(from, w0);
            if (live) {
                cnt = count_of(from, s, w0);
                off = ((uint64_t)l0.z | ((uint64_t)l0.w << 32)) & kLongOffMask;      // (an inline line's `off` is not used)
                kp = (w0 & kW0Long) ? from.arena + off : reinterpret_cast<const uint32_t *>(sp) + 2;
            }
        }
        const uint32_t meta = (uint32_t)(w0 >> 32) & 0x3FFFFu;
        // a LONG key keeps its place in the arena (both tables share it): re-inserting it through arena_alloc doubled
        // the arena's fill with every growth and the old copies were never reclaimed
        const uint32_t strip = (from.key_tag && !(w0 & kW0Long) && (meta >> 16) == kKindBytes) ? ~kTagBits : 0xFFFFFFFFu;
        table_add(to, bt, MemKey{kp, 1ull, strip}, meta >> 16, meta & 0xFFFFu, cnt, live, (live && (w0 & kW0Long)) ? off : kNoOff);
    }
    __syncthreads();
    block_table_end(bt, to);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int grid_for(uint64_t n, int per_cu) {
    uint64_t g = (n + kBlock - 1) / kBlock;
    const uint64_t cap = 256ull * per_cu;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// Workgroups of a kernel that a CU really holds (registers, LDS, wave slots): the persistent kernels below size their grids
// by it.  A grid that asks for more per CU than fit runs its surplus workgroups as a second round on a half-empty chip -
// with fixed per-CU counts the 300 bp instances did (k_finish<20> holds 189 registers: two workgroups a CU, asked for three;
// k_scan<20> 132: three, asked for five).
// VFIND_GRID_BY_OCCUPANCY: bit 0 k_scan and the filters, bit 1 k_finish, bit 2 k_dpall (A/B; default below)
static int grid_fit_mask() {
    static const int m = [] { const char *v = getenv("VFIND_GRID_BY_OCCUPANCY"); return v ? atoi(v) : 3; }();
    return m;
}
template <class K> static int occupancy_of(K kernel, size_t dyn_lds) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlock, dyn_lds) != hipSuccess) { (void)hipGetLastError(); nb = 1; }
    return nb < 1 ? 1 : nb;
}
static int w_index(int W) { return W == 10 ? 0 : W == 20 ? 1 : 2; }
template <int W> static int scan_occupancy() {
    // the list segments k_scan writes are read by one workgroup each of the filter kernels: the same count has to suit them
    int o = occupancy_of(k_scan<W>, 0);
    o = std::min(o, occupancy_of(k_filter<W>, 0));
    o = std::min(o, std::min(occupancy_of(k_bmfilter<W, 2>, 0), std::min(occupancy_of(k_bmfilter<W, 3>, 0), occupancy_of(k_bmfilter<W, 4>, 0))));
    return o;
}
static int scan_blocks_per_cu(int W) {
    // workgroups per CU of k_scan (and list segments, hence workgroups of k_filter): both kernels are persistent, and at
    // eight per CU they hold every wave slot while the other batch lane's k_finish and the table clear want in;
    // five measured best (10.8 against 10.5 G reads/s at eight, 10.7 at four, 10.4 at three, interleaved runs)
    static const int per_cu = [] { const char *v = getenv("VFIND_SCAN_BLOCKS_PER_CU"); const int x = v ? atoi(v) : 5; return x >= 1 && x <= 8 ? x : 5; }();
    const bool fit = (grid_fit_mask() & 1) != 0;
    static int occ[3] = {0, 0, 0};           // (benign race: every thread computes the same value)
    int &o = occ[w_index(W)];
    if (o == 0) o = W == 10 ? scan_occupancy<10>() : W == 20 ? scan_occupancy<20>() : scan_occupancy<40>();
    return fit ? std::min(per_cu, o) : per_cu;
}
void scan_geometry(uint64_t n, int W, uint32_t *grid, uint32_t *seg) {
    const uint32_t g = (uint32_t)grid_for(n, scan_blocks_per_cu(W));
    const uint64_t per_block = (n + (uint64_t)g * kBlock - 1) / ((uint64_t)g * kBlock) * kBlock;
    *grid = g;
    *seg = (uint32_t)per_block;
}
// list entries a batch of n reads needs whatever its row width
uint64_t scan_list_entries(uint64_t n) {
    uint64_t e = 0;
    for (int W : {10, 20, 40}) { uint32_t g, seg; scan_geometry(n, W, &g, &seg); e = std::max<uint64_t>(e, (uint64_t)g * seg); }
    return e;
}

static bool g_lut_ready[64];
static hipError_t ensure_lut() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 64 && g_lut_ready[dev]) return hipSuccess;
    static const char std_code[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
    uint8_t le[64];
    for (int b0 = 0; b0 < 4; ++b0)
        for (int b1 = 0; b1 < 4; ++b1)
            for (int b2 = 0; b2 < 4; ++b2) le[b0 + 4 * b1 + 16 * b2] = (uint8_t)std_code[b0 * 16 + b1 * 4 + b2];
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_codon_le), le, 64);
    if (e == hipSuccess && dev < 64) g_lut_ready[dev] = true;
    return e;
}

hipError_t launch_scan(hipStream_t s, int W, const uint32_t *rows, const uint16_t *lens, uint64_t n, const ScanParams &sp,
                       short2 *spans, BatchLists bl) {
    if (n == 0) return hipSuccess;
    uint32_t g, seg;
    scan_geometry(n, W, &g, &seg);
    if (bl.grid != g || bl.seg != seg) return hipErrorInvalidValue;
    switch (W) {
    case 10: hipLaunchKernelGGL(k_scan<10>, dim3(g), dim3(kBlock), 0, s, rows, lens, n, sp, spans, bl); break;
    case 20: hipLaunchKernelGGL(k_scan<20>, dim3(g), dim3(kBlock), 0, s, rows, lens, n, sp, spans, bl); break;
    case 40: hipLaunchKernelGGL(k_scan<40>, dim3(g), dim3(kBlock), 0, s, rows, lens, n, sp, spans, bl); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int M>
static void launch_dp_m(hipStream_t s, int which, int g, int W, const DpParams &dp,
                        short2 *spans, BatchLists bl, const uint32_t *list, const uint32_t *cnt, uint32_t count_stat,
                        uint64_t *counters) {
    if (which == 0) hipLaunchKernelGGL(k_dp<M>, dim3(g), dim3(kBlock), 0, s, W, dp, spans, bl, list, cnt, count_stat, counters);
    else {
        // (the class split inside the kernel works for any grid: no more workgroups than the chip holds at once)
        const bool fit = (grid_fit_mask() & 4) != 0;
        static const int occ = occupancy_of(k_dpall<M>, 0);
        const int ga = fit ? std::min(g, 256 * occ) : g;
        hipLaunchKernelGGL(k_dpall<M>, dim3(ga), dim3(kBlock), 0, s, W, dp, spans, bl, counters);
    }
}

static hipError_t launch_dp_any(hipStream_t s, int which, int W, const DpParams &dp,
                                short2 *spans, BatchLists bl, const uint32_t *list, const uint32_t *cnt, uint32_t count_stat,
                                uint64_t *counters) {
    if (bl.grid == 0) return hipSuccess;
    if (!bl.rows || !bl.lens || (W != 10 && W != 20 && W != 40)) return hipErrorInvalidValue;
    static const uint32_t g_max = [] { const char *v = getenv("VFIND_DP_GRID"); const int x = v ? atoi(v) : 768; return (uint32_t)(x >= 64 && x <= (int)kMaxGrid ? x : 768); }();      // (three workgroups a CU: round 4, same box, three rounds: cfg 2 0.787-0.790 against 0.799-0.807 ms at 1024, cfg-5 shape -3 %, cfg-4 shape level; 512 and below lose)
    const int g = which == 0 ? (int)(bl.grid < g_max ? bl.grid : g_max) : (int)g_max;
#define VF_DP_CASE(M) case M: launch_dp_m<M>(s, which, g, W, dp, spans, bl, list, cnt, count_stat, counters); break;
    switch (dp.m) {
        VF_DP_CASE(1) VF_DP_CASE(2) VF_DP_CASE(3) VF_DP_CASE(4) VF_DP_CASE(5) VF_DP_CASE(6) VF_DP_CASE(7) VF_DP_CASE(8)
        VF_DP_CASE(9) VF_DP_CASE(10) VF_DP_CASE(11) VF_DP_CASE(12) VF_DP_CASE(13) VF_DP_CASE(14) VF_DP_CASE(15) VF_DP_CASE(16)
        VF_DP_CASE(17) VF_DP_CASE(18) VF_DP_CASE(19) VF_DP_CASE(20) VF_DP_CASE(21) VF_DP_CASE(22) VF_DP_CASE(23) VF_DP_CASE(24)
        VF_DP_CASE(25) VF_DP_CASE(26) VF_DP_CASE(27) VF_DP_CASE(28) VF_DP_CASE(29) VF_DP_CASE(30) VF_DP_CASE(31) VF_DP_CASE(32)
    default: return hipErrorInvalidValue;
    }
#undef VF_DP_CASE
    return hipGetLastError();
}

// full-width DP over the role's own list (scoring schemes that give the filter no usable bound)
hipError_t launch_dp(hipStream_t s, int W, const DpParams &dp, short2 *spans,
                     BatchLists bl, uint64_t *counters) {
    const uint32_t *list = dp.is_prefix ? bl.listP : bl.listS;
    const uint32_t *cnt = dp.is_prefix ? bl.cntP : bl.cntS;
    return launch_dp_any(s, 0, W, dp, spans, bl, list, cnt, 1u, counters);
}

// every task class of the filter's output, one launch
hipError_t launch_dp_classes(hipStream_t s, int W, const DpParams &dp, short2 *spans,
                             BatchLists bl, uint64_t *counters) {
    // (the banded class as a launch of its own, 77 instead of 113 registers per wave, measured the same: 9.48 vs 9.43 G reads/s)
    return launch_dp_any(s, 1, W, dp, spans, bl, nullptr, nullptr, 0u, counters);
}

hipError_t launch_filter(hipStream_t s, int W, const DpParams &dp, short2 *spans,
                         BatchLists bl, uint64_t *counters) {
    if (bl.grid == 0) return hipSuccess;
    if (!bl.rows || !bl.lens) return hipErrorInvalidValue;
    const int g = (int)bl.grid;                       // one workgroup per list segment (<= kMaxGrid)
    switch (W) {
    case 10: hipLaunchKernelGGL(k_filter<10>, dim3(g), dim3(kBlock), 0, s, dp, spans, bl, counters); break;
    case 20: hipLaunchKernelGGL(k_filter<20>, dim3(g), dim3(kBlock), 0, s, dp, spans, bl, counters); break;
    case 40: hipLaunchKernelGGL(k_filter<40>, dim3(g), dim3(kBlock), 0, s, dp, spans, bl, counters); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int W>
static void launch_bmfilter_w(hipStream_t s, int g, const DpParams &dp, short2 *spans, BatchLists bl,
                              uint64_t *counters) {
    switch (dp.nblk) {
    case 2: hipLaunchKernelGGL((k_bmfilter<W, 2>), dim3(g), dim3(kBlock), 0, s, dp, spans, bl, counters); break;
    case 3: hipLaunchKernelGGL((k_bmfilter<W, 3>), dim3(g), dim3(kBlock), 0, s, dp, spans, bl, counters); break;
    default: hipLaunchKernelGGL((k_bmfilter<W, 4>), dim3(g), dim3(kBlock), 0, s, dp, spans, bl, counters); break;
    }
}
hipError_t launch_bmfilter(hipStream_t s, int W, const DpParams &dp, short2 *spans, BatchLists bl,
                           uint64_t *counters) {
    if (bl.grid == 0) return hipSuccess;
    if (!bl.rows || !bl.lens) return hipErrorInvalidValue;
    if (dp.nblk < 2 || dp.nblk > 4) return hipErrorInvalidValue;
    const int g = (int)bl.grid;
    switch (W) {
    case 10: launch_bmfilter_w<10>(s, g, dp, spans, bl, counters); break;
    case 20: launch_bmfilter_w<20>(s, g, dp, spans, bl, counters); break;
    case 40: launch_bmfilter_w<40>(s, g, dp, spans, bl, counters); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}


// workgroups per CU of k_finish.  Three measured best next to a second batch lane (10.5 vs 10.3 at two and 10.1 G reads/s
// at four, interleaved runs): the kernel is bound by the memory side (slot lines, CAS), more resident waves add
// contention there and take registers from the DP waves of the other lane
static bool finish_u2();
static int finish_per_cu(int W, bool dna) {
    static const int per_cu = [] { const char *v = getenv("VFIND_FINISH_BLOCKS_PER_CU"); const int x = v ? atoi(v) : 3; return x >= 1 && x <= 8 ? x : 3; }();
    const bool fit = (grid_fit_mask() & 2) != 0;
    static int occ[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    int &o = occ[w_index(W)][dna ? 1 : 0];
    if (o == 0) {
        const size_t dyn = 2 * kLogMaxParts * sizeof(uint32_t);
        if (W == 10) o = dna ? (finish_u2() ? occupancy_of(k_finish<10, 2, true>, dyn) : occupancy_of(k_finish<10, 1, true>, dyn))
                             : (finish_u2() ? occupancy_of(k_finish<10, 2, false>, dyn) : occupancy_of(k_finish<10, 1, false>, dyn));
        else if (W == 20) o = dna ? occupancy_of(k_finish<20, 1, true>, dyn) : occupancy_of(k_finish<20, 1, false>, dyn);
        else o = dna ? occupancy_of(k_finish<40, 1, true>, dyn) : occupancy_of(k_finish<40, 1, false>, dyn);
    }
    return fit ? std::min(per_cu, o) : per_cu;
}
static bool finish_u2() {
    static const bool u2 = [] { const char *v = getenv("VFIND_FINISH_U"); return v && atoi(v) == 2; }();
    return u2;
}
// what a k_finish launch over n reads needs of the count log: workgroups, entries per wave segment, entries in all
uint64_t count_log_entries(uint64_t n, int W, bool dna, uint32_t *grid, uint32_t *cap_one) {
    const uint32_t g = (uint32_t)grid_for(n, finish_per_cu(W, dna));
    const uint64_t gstride = (uint64_t)g * kBlock;
    const uint64_t per_wave = (n + gstride - 1) / gstride * WAVE;      // a wave sees this many reads (U reads a lane and trip: same total)
    const uint64_t cap = (per_wave + 2 * WAVE + 15) / 16 * 16;          // (U = 2 rounds the trips up to pairs)
    *grid = g;
    *cap_one = (uint32_t)cap;
    return cap * g * (kBlock / WAVE);
}
// log entries and (slot, count) blocks a batch of n reads needs whatever its row width and key kind
uint64_t count_log_entries_max(uint64_t n, uint32_t *grid_max) {
    uint64_t e = 0;
    uint32_t gm = 0;
    for (int W : {10, 20, 40})
        for (int dna = 0; dna < 2; ++dna) {
            uint32_t g, cap;
            e = std::max(e, count_log_entries(n, W, dna != 0, &g, &cap));
            gm = std::max(gm, g);
        }
    *grid_max = gm;
    return e;
}
// slot parts of the count log for a table of `slots` slots and a batch of n reads: parts = 0 -> k_log_direct
static void count_log_parts(uint64_t slots, uint64_t n, uint32_t *parts, uint32_t *shift) {
    static const int force = [] { const char *v = getenv("VFIND_COUNT_LOG"); return v ? atoi(v) : 0; }();   // 1 direct, 2 by parts (A/B runs, tests)
    // by parts pays when the batch touches most ranges: a workgroup per range zeroes and walks 32 KiB of LDS whether
    // the range got entries or not, and every part is read by all of its ranges
    bool by_parts = slots <= (1ull << 26) && n >= slots / 64;
    if (force == 1) by_parts = false;
    if (force == 2 && slots <= (1ull << 31)) by_parts = true;
    if (!by_parts) { *parts = 0; *shift = 0; return; }
    uint32_t sh = 0;
    while ((slots >> sh) > kLogMaxParts) ++sh;                          // parts = slots >> shift <= kLogMaxParts, a power of two
    const uint32_t range_sh = 13;                                       // kLogRange = 2^13: a range never straddles two parts
    static_assert(kLogRange == 1u << 13, "range shift");
    if (sh < range_sh && slots > kLogRange) sh = range_sh;
    if (slots <= kLogRange) sh = 31;                                    // one range, one part
    *shift = sh;
    *parts = (uint32_t)((slots - 1) >> sh) + 1u;
}

hipError_t launch_finish(hipStream_t s, int W, const uint32_t *rows, const uint16_t *lens, uint64_t n, const short2 *spans,
                         bool skip_translation, Table t, CountLog log) {
    if (n == 0) return hipSuccess;
    if (n >> 31) return hipErrorInvalidValue;         // read indices are 31-bit in the work lists (bit 31 of an entry is a flag)
    hipError_t e = ensure_lut();
    if (e != hipSuccess) return e;
    // one read per lane and trip measured best once the kernel runs two workgroups per CU (7.44 vs 7.23 G reads/s over
    // four interleaved runs each): the one-read variant needs 19 registers fewer (125 against 144)
    const bool u2 = finish_u2();
    uint32_t g_log, cap_one;
    const uint64_t need = count_log_entries(n, W, skip_translation, &g_log, &cap_one);
    const int g = (int)g_log;
    if (log.cap_total < need || !log.one || !log.n_one || !log.many || !log.n_many) return hipErrorInvalidValue;
    log.cap_one = cap_one;
    log.grid = g_log;
    count_log_parts((uint64_t)t.slot_mask + 1, n, &log.parts, &log.shift);
    // (tot[2][..] and cur[2][..] are zero: the caller resets them when it starts the batch, off the table kernels' path)
    if (log.parts && (!log.tot || !log.cur || !log.one_by_part || !log.many_by_part)) return hipErrorInvalidValue;
    const size_t dyn = log.parts ? 2 * kLogMaxParts * sizeof(uint32_t) : 0;
#define VF_FINISH(WW, UU)                                                                                              \
    do {                                                                                                               \
        if (skip_translation) hipLaunchKernelGGL((k_finish<WW, UU, true>), dim3(g), dim3(kBlock), dyn, s, rows, lens, n, spans, t, log); \
        else hipLaunchKernelGGL((k_finish<WW, UU, false>), dim3(g), dim3(kBlock), dyn, s, rows, lens, n, spans, t, log);               \
    } while (0)
    switch (W) {
    case 10: {
#ifdef VF_FINISH_ABLATION      // diagnosis build only (make ablation): these variants give wrong counts by construction
        static const int dbg = [] { const char *v = getenv("VFIND_FINISH_DBG"); return v ? atoi(v) : 0; }();
        if (dbg == 9) return hipSuccess;          // no table kernel at all: what the chains of two lanes do on their own
        if (dbg == 1) { hipLaunchKernelGGL((k_finish<10, 2, false, 1>), dim3(g), dim3(kBlock), dyn, s, rows, lens, n, spans, t, log); break; }
        if (dbg == 2) { hipLaunchKernelGGL((k_finish<10, 2, false, 2>), dim3(g), dim3(kBlock), dyn, s, rows, lens, n, spans, t, log); break; }
        if (dbg == 3) { hipLaunchKernelGGL((k_finish<10, 2, false, 3>), dim3(g), dim3(kBlock), dyn, s, rows, lens, n, spans, t, log); break; }
        if (dbg == 4) { hipLaunchKernelGGL((k_finish<10, 2, false, 4>), dim3(g), dim3(kBlock), dyn, s, rows, lens, n, spans, t, log); break; }
#endif
        if (u2) VF_FINISH(10, 2); else VF_FINISH(10, 1);
        break;
    }
    case 20: VF_FINISH(20, 1); break;
    case 40: VF_FINISH(40, 1); break;
    default: return hipErrorInvalidValue;
    }
#undef VF_FINISH
    return hipGetLastError();
}

// the count log of the k_finish launch over the same n reads, applied to the table's counts (nothing in k_finish reads a
// count, so the next batch's k_finish may run beside this)
hipError_t launch_count_log(hipStream_t s, int W, bool skip_translation, uint64_t n, Table t, CountLog log) {
    if (n == 0) return hipSuccess;
    uint32_t g_log, cap_one;
    (void)count_log_entries(n, W, skip_translation, &g_log, &cap_one);
    const int g = (int)g_log;
    log.cap_one = cap_one;
    log.grid = g_log;
    count_log_parts((uint64_t)t.slot_mask + 1, n, &log.parts, &log.shift);
#ifdef VF_FINISH_ABLATION
    { static const int dbg = [] { const char *v = getenv("VFIND_FINISH_DBG"); return v ? atoi(v) : 0; }(); if (dbg == 9) return hipSuccess; }
#endif
    if (log.parts) {
        hipLaunchKernelGGL(k_log_scatter, dim3(g), dim3(kBlock), 0, s, log, t.counters);
        const uint64_t nslots = (uint64_t)t.slot_mask + 1;
        const uint64_t ranges = nslots <= kLogRange ? 1 : nslots / kLogRange;
        hipLaunchKernelGGL(k_log_apply, dim3((unsigned)std::min<uint64_t>(ranges, kMaxGrid)), dim3(kBlock), 0, s, t, log);
    } else {
        hipLaunchKernelGGL(k_log_direct, dim3(g), dim3(kBlock), 0, s, t, log);
    }
    return hipGetLastError();
}

hipError_t launch_generic(hipStream_t s, const uint8_t *bytes, const uint64_t *offs, uint64_t n, const GenericParams &gp,
                          int32_t *starts, int32_t *ends, int32_t *ws, uint64_t ws_stride, uint32_t ws_rows, Table t) {
    if (n == 0) return hipSuccess;
    hipError_t e = ensure_lut();
    if (e != hipSuccess) return e;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g * kBlock > ws_stride) g = ws_stride / kBlock;
    if (g < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_generic, dim3((unsigned)g), dim3(kBlock), 0, s, bytes, offs, n, gp, starts, ends, ws, ws_stride, ws_rows, t,
                       (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    return hipGetLastError();
}

// reads listed by the device framer; their number is only known on the device, so the grid is fixed
hipError_t launch_generic_indexed(hipStream_t s, const uint8_t *text, const uint32_t *idx_start, const uint32_t *idx_len, const uint32_t *n_dev,
                                  const GenericParams &gp, int32_t *ws, uint64_t ws_stride, uint32_t ws_rows, Table t) {
    hipError_t e = ensure_lut();
    if (e != hipSuccess) return e;
    const uint64_t g = ws_stride / kBlock;
    if (g < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_generic, dim3((unsigned)g), dim3(kBlock), 0, s, text, (const uint64_t *)nullptr, (uint64_t)0, gp, (int32_t *)nullptr, (int32_t *)nullptr,
                       ws, ws_stride, ws_rows, t, idx_start, idx_len, n_dev);
    return hipGetLastError();
}

// newline index of a text chunk: line_end[0..lines), info = {lines, longest line, first error, generic reads}
hipError_t launch_line_index(hipStream_t s, const uint8_t *text, uint64_t n, uint32_t *tile_cnt, uint32_t *tile_off, uint32_t *line_end, uint32_t *info) {
    if (n == 0) return hipSuccess;
    const uint32_t tiles = (uint32_t)((n + kTileBytes - 1) / kTileBytes);
    hipLaunchKernelGGL(k_nl_count, dim3(tiles), dim3(kBlock), 0, s, text, n, tile_cnt);
    hipLaunchKernelGGL(k_nl_scan, dim3(1), dim3(1024), 0, s, tile_cnt, tiles, tile_off, info);
    hipLaunchKernelGGL(k_nl_write, dim3(tiles), dim3(kBlock), 0, s, text, n, tile_off, line_end);
    hipLaunchKernelGGL(k_line_max, dim3(256), dim3(kBlock), 0, s, line_end, info, info);
    return hipGetLastError();
}

hipError_t launch_frame_pack(hipStream_t s, int W, const uint8_t *text, const uint32_t *line_end, uint32_t n_rec, uint32_t *rows, uint16_t *lens,
                             uint32_t *gen_start, uint32_t *gen_len, uint32_t *info, bool all_generic) {
    if (n_rec == 0) return hipSuccess;
    const int g = (int)std::min<uint32_t>((n_rec + kBlock - 1) / kBlock, 4096u);
    switch (W) {
    case 10: hipLaunchKernelGGL(k_frame_pack<10>, dim3(g), dim3(kBlock), 0, s, text, line_end, n_rec, rows, lens, gen_start, gen_len, info, all_generic ? 1 : 0); break;
    case 20: hipLaunchKernelGGL(k_frame_pack<20>, dim3(g), dim3(kBlock), 0, s, text, line_end, n_rec, rows, lens, gen_start, gen_len, info, all_generic ? 1 : 0); break;
    case 40: hipLaunchKernelGGL(k_frame_pack<40>, dim3(g), dim3(kBlock), 0, s, text, line_end, n_rec, rows, lens, gen_start, gen_len, info, all_generic ? 1 : 0); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// one launch zeroes the table's four regions with 16-byte stores from every CU (the runtime's fill kernel runs
// 256 workgroups per memset and took 0.19 ms per clear for a 2^21-slot table: serial time between two k_finish)
__global__ __launch_bounds__(kBlock) void k_clear(ClearRegions cr) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint4 *p = reinterpret_cast<uint4 *>(cr.p[q]);
        const uint64_t n16 = cr.bytes[q] >> 4;
        for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * kBlock) p[i] = z;
        if (blockIdx.x == 0 && threadIdx.x < ((cr.bytes[q] & 15u) >> 2)) reinterpret_cast<uint32_t *>(p + n16)[threadIdx.x] = 0u;
    }
}

#ifdef VF_FINISH_ABLATION
extern "C" int vf_debug_fin_trace(uint64_t *out) {         // diagnosis build: kMaxGrid x 4 waves x 12 words
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fin_trace), sizeof(uint64_t) * kMaxGrid * (kBlock / WAVE) * 12) == hipSuccess ? 0 : 4;
}
extern "C" int vf_debug_dp_trace(uint64_t *out) {          // diagnosis build: 2 x kMaxGrid x 4 words
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dp_trace), sizeof(uint64_t) * 2 * kMaxGrid * 4) == hipSuccess ? 0 : 4;
}
#endif

// A few words of device memory written straight into page-locked host memory by a one-wave kernel.  An 8-byte
// hipMemcpyAsync device-to-host goes through the copy engine's queue, in order behind every 47 MB host-to-device copy
// the other slots have queued: the entry count that rides back with a batch (and the batch's `done` event behind it)
// arrived seven copies late, and the line index count of a FASTQ chunk, which the host waits for, likewise.
__global__ void k_peek(const uint32_t *__restrict__ src, uint32_t *__restrict__ host_dst, uint32_t n_dwords) {
    if (threadIdx.x < n_dwords) __hip_atomic_store(host_dst + threadIdx.x, src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_peek(hipStream_t s, const void *d_src, void *h_dst, uint32_t n_dwords) {
    if (n_dwords == 0 || n_dwords > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_peek, dim3(1), dim3(64), 0, s, (const uint32_t *)d_src, (uint32_t *)h_dst, n_dwords);
    return hipGetLastError();
}

hipError_t launch_clear(hipStream_t s, const ClearRegions &cr) {
    for (int q = 0; q < 4; ++q)
        if (((uintptr_t)cr.p[q] & 15u) || (cr.bytes[q] & 3u)) return hipErrorInvalidValue;
    static const int grid = [] { const char *v = getenv("VFIND_CLEAR_GRID"); const int x = v ? atoi(v) : 2048; return x >= 1 && x <= (int)kMaxGrid ? x : 2048; }();
    hipLaunchKernelGGL(k_clear, dim3(grid), dim3(kBlock), 0, s, cr);
    return hipGetLastError();
}

hipError_t launch_export(hipStream_t s, Table t, uint64_t *out_refs, uint64_t *out_counts, uint32_t *out_arena, uint32_t G, uint32_t owner) {
    const int g = grid_for((uint64_t)t.slot_mask + 1, 8);
    hipLaunchKernelGGL(k_export, dim3(g), dim3(kBlock), 0, s, t, out_refs, out_counts, out_arena, G, owner);
    return hipGetLastError();
}

hipError_t launch_owner_count(hipStream_t s, Table t, uint32_t G, uint64_t *out) {
    if (G == 0 || G > kMaxOwners) return hipErrorInvalidValue;
    const int g = grid_for((uint64_t)t.slot_mask + 1, 8);
    hipLaunchKernelGGL(k_owner_count, dim3(g), dim3(kBlock), 0, s, t, G, reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

hipError_t launch_merge(hipStream_t s, Table t, const uint64_t *refs, const uint64_t *counts, const uint32_t *arena, uint64_t n) {
    if (n == 0) return hipSuccess;
    const int g = grid_for(n, 8);
    hipLaunchKernelGGL(k_merge, dim3(g), dim3(kBlock), 0, s, t, refs, counts, arena, n);
    return hipGetLastError();
}

hipError_t launch_rehash(hipStream_t s, Table from, Table to) {
    const int g = grid_for((uint64_t)from.slot_mask + 1, 8);
    hipLaunchKernelGGL(k_rehash, dim3(g), dim3(kBlock), 0, s, from, to);
    return hipGetLastError();
}

}  // namespace vf
