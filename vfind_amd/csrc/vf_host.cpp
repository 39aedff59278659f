// vf_host.cpp -- host runtime behind the C ABI of include/vfind_hip.h.
//
// Owns: parameter derivation (lib.rs:100-110, 236-261), gzip FASTQ decode and 2-bit packing on
// host threads into pinned staging, the per-GPU batch pipeline (hipMemcpyAsync of batch k+1
// overlapped with the kernels of batch k on per-slot streams), the device count table, and the
// result build (lib.rs:312-319).  No CPU fallback exists: every per-read decision is made by the
// kernels in vf_k_*.hip; the host only decodes, packs, launches and formats.
#include "../../include/vfind_hip.h"
#include "vf_device.h"
#include "vf_internal.h"
#include "vf_result.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

using namespace vf;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail(VF_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

extern "C" const char *vf_last_error(void) { return g_err.c_str(); }
int vf_internal_fail_noexcept(int code, const char *msg) noexcept {
    try { g_err = msg; } catch (...) { g_err.clear(); }      // (assigning a short literal does not allocate: SSO)
    return code;
}
bool vf_internal_test_throw(const char *where) {
    const char *e = getenv("VFIND_TEST_THROW");
    return e && strcmp(e, where) == 0;
}
extern "C" int vf_abi_version(void) { return VF_ABI_VERSION; }

// base code of an adapter letter (the adapter was checked to be ACGT-only before this is used)
static inline uint32_t base_code(unsigned char ch) { return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 0xFFu; }

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct vf_local_group;

namespace {

struct EventPair { hipEvent_t a, b; int kind; };
enum { K_SCAN = 0, K_FILTER, K_DPW, K_DPREST, K_FIN, K_GEN, K_LOG, K_H2D, K_FRAME, K_KINDS };

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;   // elements
    int ensure(size_t n) {
        if (n <= cap) return VF_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = std::max(n, (size_t)16);
        if (hipMalloc((void **)&p, want * sizeof(T)) != hipSuccess) return fail(VF_ENOMEM, "hipMalloc failed");
        cap = want;
        return VF_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
template <typename T> struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return VF_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = std::max(n, (size_t)16);
        if (hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault) != hipSuccess) return fail(VF_ENOMEM, "hipHostMalloc failed");
        cap = want;
        return VF_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct Workspace {               // device scratch of one in-flight batch
    DevBuf<short2> spans;
    DevBuf<uint32_t> listP, listS, listW[4], winW[4], winW2, listF, listN, winN, listR, listB, winB, cnt;
    // count log of the batch's k_finish (CountLog, vf_device.h)
    DevBuf<uint32_t> log_one, log_one_by_part, log_n;     // log_n: n_one[kMaxGrid * 4], n_many[kMaxGrid], tot[2][parts], cur[2][parts]
    DevBuf<uint2> log_many, log_many_by_part;
    // The log is applied on the context's main stream, off the batch's own: log_written (batch stream, behind k_finish) ->
    // main stream: scatter, apply, part totals zeroed again -> log_applied, which the next k_finish over this workspace
    // waits for.  Nothing in a batch's chain of kernels waits for the counts of the batch before it.
    hipEvent_t log_written = nullptr, log_applied = nullptr;
    bool log_pending = false;                             // log_applied has been recorded and not yet waited for
    // main: the stream the log is applied on (it also zeroes a new totals array: k_finish adds to it)
    int ensure_log(size_t n, hipStream_t main) {
        int rc;
        uint32_t g;
        const uint64_t e = count_log_entries_max(n, &g);
        if ((rc = log_one.ensure(e)) || (rc = log_one_by_part.ensure(e))) return rc;
        if ((rc = log_many.ensure((size_t)g * kLogMany)) || (rc = log_many_by_part.ensure((size_t)g * kLogMany))) return rc;
        if (!log_written && hipEventCreateWithFlags(&log_written, hipEventDisableTiming) != hipSuccess) return fail(VF_EDEVICE, "hipEventCreate failed");
        if (!log_applied && hipEventCreateWithFlags(&log_applied, hipEventDisableTiming) != hipSuccess) return fail(VF_EDEVICE, "hipEventCreate failed");
        if (!log_n.p) {
            if ((rc = log_n.ensure(5 * (size_t)kMaxGrid + 4 * (size_t)kLogMaxParts))) return rc;
            // (on the main stream and behind an event: a hipMemset on the null stream does not order with these non-blocking
            // streams and once landed on top of a first batch's entry counts)
            if (hipMemsetAsync(log_n.p, 0, log_n.cap * sizeof(uint32_t), main) != hipSuccess || hipEventRecord(log_applied, main) != hipSuccess)
                return fail(VF_EDEVICE, "hipMemsetAsync failed");
            log_pending = true;
        }
        return VF_OK;
    }
    CountLog count_log() const {
        CountLog l{};
        l.one = log_one.p; l.one_by_part = log_one_by_part.p; l.many = log_many.p; l.many_by_part = log_many_by_part.p;
        l.n_one = log_n.p; l.n_many = log_n.p + 4 * (size_t)kMaxGrid;
        l.tot = log_n.p + 5 * (size_t)kMaxGrid; l.cur = l.tot + 2 * (size_t)kLogMaxParts;
        l.cap_total = log_one.cap;
        return l;
    }
    int ensure(size_t n, int /*W*/) {
        int rc;
        const size_t e = (size_t)scan_list_entries(n);
        if ((rc = spans.ensure(n))) return rc;
        if ((rc = listP.ensure(e)) || (rc = listS.ensure(e)) || (rc = winW2.ensure(e)) || (rc = listF.ensure(e))) return rc;
        if ((rc = listN.ensure(e)) || (rc = winN.ensure(e)) || (rc = listR.ensure(e)) || (rc = listB.ensure(e)) || (rc = winB.ensure(e))) return rc;
        for (int q = 0; q < 4; ++q) if ((rc = listW[q].ensure(e)) || (rc = winW[q].ensure(e))) return rc;
        return cnt.ensure(10 * kMaxGrid + 16);
    }
    void release() {
        spans.release(); listP.release(); listS.release(); listF.release(); winW2.release(); cnt.release();
        listN.release(); winN.release(); listR.release(); listB.release(); winB.release();
        log_one.release(); log_one_by_part.release(); log_n.release(); log_many.release(); log_many_by_part.release();
        if (log_written) (void)hipEventDestroy(log_written);
        if (log_applied) (void)hipEventDestroy(log_applied);
        log_written = log_applied = nullptr; log_pending = false;
        for (int q = 0; q < 4; ++q) { listW[q].release(); winW[q].release(); }
    }
};

struct Slot;
struct ResLane {                 // one of the in-order lanes vf_process_device_packed alternates between
    PinBuf<uint64_t> h_entries;  // table entry count as of this lane's latest finished batch
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    std::atomic<bool> in_flight{false};
    std::atomic<uint64_t> inflight_reads{0};   // reads of the batch `done` stands for (capacity_refresh)
    uint32_t chunk_row = 0;
    Workspace ws;
};

struct Slot {
    PinBuf<uint64_t> h_entries;  // table entry count as of this slot's latest finished batch
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    uint32_t index = 0;          // chunk-state row of this slot's stream
    bool busy = false;           // owned by a submitting thread
    std::atomic<bool> in_flight{false};      // event recorded, not yet waited
    std::atomic<uint64_t> inflight_reads{0};  // reads of the batch `done` stands for (capacity_refresh)
    std::atomic<bool> cap_pending{false};     // those reads still count as possible new keys (capacity_retire takes them off)
    std::atomic<uint64_t> cap_seq{0};         // batches noted on this slot: a retire decided on an earlier batch's event does not touch a later batch
    // fast path staging
    PinBuf<uint32_t> h_rows; PinBuf<uint16_t> h_lens;
    DevBuf<uint32_t> d_rows; DevBuf<uint16_t> d_lens;
    Workspace ws;
    // generic path staging
    PinBuf<uint8_t> h_gbytes; PinBuf<uint64_t> h_goffs;
    DevBuf<uint8_t> d_gbytes; DevBuf<uint64_t> d_goffs;
    DevBuf<int32_t> d_gstart, d_gend, d_gws;
    PinBuf<int32_t> h_gstart, h_gend;
    PinBuf<short2> h_spans;
    std::vector<uint32_t> fast_idx, gen_idx;   // original positions (vf_spans_ascii)
    // device framing of FASTQ text (vf_internal_submit_text)
    PinBuf<uint8_t> h_text; DevBuf<uint8_t> d_text;
    DevBuf<uint32_t> d_tile_cnt, d_tile_off, d_line_end, d_info, d_gen_start, d_gen_len;
    PinBuf<uint32_t> h_info;
    bool text_pending = false;   // h_info of the latest text batch has not been looked at yet
    uint64_t last_use = 0;       // ticket of the latest acquire (acquire_slot takes the free slot with the oldest one)
    uint64_t batch_cap = 0;      // reads the staging was acquired for (vf_batch_acquire), 0 = not handed out
    uint32_t batch_W = 0;
};

}  // namespace

struct vf_ctx {
    int dev = 0;
    // derived parameters
    std::string pre, suf;
    int32_t match = 3, mismatch = -2, gopen = 5, gext = 2;
    bool skip_pre = false, skip_suf = false;   // skip alignment (accept == 1.0)
    int64_t T_pre = 0, T_suf = 0;
    bool skip_translation = false;
    bool fast_ok = false;
    bool filter_on = false;      // Myers pre-DP filter + windowed DP usable for these scores
    bool block_filter = true;    // block filter in front of it (VFIND_BLOCK_FILTER=0 switches it off: A/B runs)
    uint32_t n_threads = 3, queue_len = 2, batch_reads = 1u << 18;
    ScanParams sp{};
    DpParams dpP{}, dpS{};
    GenericParams gp{};
    DevBuf<uint8_t> d_adapters;
    // table
    Table tab{};
    uint64_t table_slots = 0, arena_dwords = 0;
    DevBuf<uint4> t_slots; DevBuf<unsigned long long> t_counts; DevBuf<uint32_t> t_arena; DevBuf<uint64_t> t_counters;
    DevBuf<unsigned long long> t_chunks;
    // The spare table: vf_table_clear starts the clear of the table in use and switches to the spare, which the
    // previous clear left empty - the batches submitted next count at once, beside the clear, instead of queueing behind
    // it (one table: finish -> clear -> finish is the serial spine of every step of a run of short jobs).  The table
    // kernels of the two batch lanes are then ordered by an event of their own (fin_event): side by side they take 40 %
    // longer each and the step gets slower, one after the other with the clear beside them it gets 3 % faster.
    // Made at the first clear (a context that is never cleared never pays for it); VFIND_TABLE_SPARE=0 keeps one table.
    struct Spare {
        DevBuf<uint4> slots; DevBuf<unsigned long long> counts; DevBuf<uint32_t> arena; DevBuf<uint64_t> counters;
        DevBuf<unsigned long long> chunks;
        Table tab{};
        uint64_t table_slots = 0, arena_dwords = 0;
        hipEvent_t cleared = nullptr;
        bool valid = false;
    } spare;
    bool spare_on = true;        // VFIND_TABLE_SPARE=0: one table, cleared in place
    bool epoch_tags = false;     // letter-key table (no skip_translation): cleared by epoch (VFIND_EPOCH_TAGS=0: always by zeroing)
    // pipeline
    std::vector<std::unique_ptr<Slot>> slots;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t slot_ticket = 0;
    hipStream_t main_stream = nullptr;
    hipEvent_t clear_event = nullptr;   // recorded on main_stream after each table clear
    hipEvent_t fin_event = nullptr;     // spare table: recorded after each table kernel, the next one waits for it (one at a time)
    std::mutex log_mu;                  // a batch's count log is applied on the main stream (run_packed)
    std::atomic<bool> fin_recorded{false};
    std::mutex fin_mu;                  // wait + launch + record of the table kernel: one submitter at a time, so the kernels really chain
    static constexpr int kMaxResLanes = 4;
    ResLane res[kMaxResLanes];   // vf_process_device_packed: batch k+1's scan/DP overlaps batch k's finish
    uint32_t n_res = 2, res_next = 0;
    DevBuf<uint64_t> exp_refs, exp_counts;
    // multi-GPU exchange
    DevBuf<uint64_t> d_part;            // k_owner_count output
    uint64_t part[2 * 64] = {};         // entries, key dwords per owner of the latest partition_counts
    std::atomic<uint32_t> part_world{0};   // 0 = stale (the table changed since)
    void *comm = nullptr;               // ncclComm_t of vf_comm_init
    vf_local_group *local_group = nullptr;   // or the in-process group of vf_comm_init_local (not owned)
    int comm_rank = 0, comm_world = 1;
    PinBuf<uint64_t> h_hdr;             // header of a caller-owned blob, read in stream order (read_blob_header)
    DevBuf<uint64_t> d_comm_words;      // status / size words of the exchange's all-gathers, allocated with the communicator
    // stats / profiling
    std::atomic<uint64_t> n_reads{0}, n_generic{0};
    std::atomic<int> profiling{0};      // 0 off, 1 every kernel bracket, 2 k_scan and k_finish only, 3 the same on every 3rd batch
    std::atomic<uint32_t> prof_batch{0};   // batches seen by run_packed (mode 3 samples on it; the decision is per call, Timed::sampled)
    std::mutex ev_mu;
    std::vector<EventPair> pending_events;
    double ms_result = 0;               // wall time of the latest vf_finish (export + device result build + copy back)
    uint64_t result_rows = 0;
    uint64_t dup_rows = 0;              // entries the latest result build folded into another row (vf_stats.dup_rows)
    VfFastqStats fq;                    // stage timers of vf_run_fastq
    double ms[K_KINDS] = {};
    uint64_t launches[K_KINDS] = {};
    // async error from worker threads
    // table growth: submitters hold it shared while they enqueue, table_grow exclusively
    std::shared_mutex table_mu;
    std::mutex cap_mu;                          // guards the two counters below
    uint64_t upper_entries = 0;                 // entries at the last exact count + every read submitted since
    // the same bound without the drain: the largest entry count a FINISHED batch has brought back (the counter only grows, so
    // the batch that finished last brought one that covers every finished batch) + the reads of the batches that have not
    // been seen to finish (each may still add a key)
    uint64_t known_entries = 0, pending_reads = 0;
    uint64_t unenqueued = 0;                    // reads of submitters between their capacity check and their enqueue
    uint64_t upper_arena = 0, unenqueued_arena = 0;   // the same pair for key-arena dwords
    std::mutex err_mu;
    int async_rc = 0;
    std::string async_msg;
};


// ------------------------------------------------------------------------------------------------
// parameter derivation
// ------------------------------------------------------------------------------------------------
// align_threshold_preflight (lib.rs:100-110): Ok(false) for (0,1), Ok(true) for 1, else ValueError
static int preflight(double t, bool *skip) {
    if (t > 0. && t < 1.) { *skip = false; return VF_OK; }
    if (t == 1.) { *skip = true; return VF_OK; }
    return fail(VF_EINVAL, "Accept alignment threshold must be between 0 and 1.");
}
// lib.rs:260-261 then the strict '>' of lib.rs:157 as an integer bound
static int64_t int_threshold(double accept, int32_t match, size_t len) {
    const double min_score = accept * (double)match * (double)len;   // left to right
    const double f = std::floor(min_score);
    if (f >= 9.0e15) return INT64_MAX;
    if (f <= -9.0e15) return INT64_MIN + 1;
    return (int64_t)f + 1;
}

static bool pure_acgt(const std::string &s) {
    for (unsigned char c : s) if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return false;
    return true;
}
static uint64_t pack_adapter(const std::string &s) {
    uint64_t v = 0;
    for (size_t i = 0; i < s.size(); ++i) v |= (uint64_t)(base_code((unsigned char)s[i]) & 3u) << (2 * i);
    return v;
}
// Largest number of unit edits (mismatch, read base vs gap, adapter base vs gap, unaligned adapter end
// base) an alignment with score >= T can contain.  Its loss against m matches is at most
// L = match*m - T; a mismatch costs match-mismatch, an unaligned end base match, an adapter base in a
// gap at least min(open,ext)+match, a read base in a gap `open` for the first of a gap and `ext` for
// the rest.  Returns -1 when nothing can reach T, INT32_MAX when the scheme gives no bound.
static int32_t max_unit_errors(int64_t T, size_t m, int32_t match, int32_t mismatch, int32_t gopen, int32_t gext) {
    const int64_t L = (int64_t)match * (int64_t)m - T;
    if (L < 0) return -1;
    if (match < 1 || match - mismatch < 1 || gext < 1 || gopen < 0) return INT32_MAX;
    const int64_t c_other = std::min<int64_t>(std::min<int64_t>(match, match - mismatch), std::min(gopen, gext) + (int64_t)match);
    int64_t best = L / c_other;
    for (int64_t g = 1; g * gopen <= L && g <= 4096; ++g)            // g read-side gaps, total length len
        for (int64_t len = g; g * gopen + (len - g) * gext <= L; ++len)
            best = std::max(best, len + (L - g * gopen - (len - g) * gext) / c_other);
    return best > 1000000 ? INT32_MAX : (int32_t)best;
}

// The same bound in INDEL units (a mismatch counts two: a deletion and an insertion), for the indel scan of the pre-DP filter:
// the most such units an alignment with score >= T can contain.  A mismatch loses match-mismatch for 2 units, any other single
// unit (unaligned adapter end base, adapter base in a gap) at least min(match, min(open,ext)+match), read bases in g gaps of
// total length len lose g*open + (len-g)*ext for len units.  -1: nothing reaches T; INT32_MAX: no (useful) bound.
static int32_t max_indel_units(int64_t T, size_t m, int32_t match, int32_t mismatch, int32_t gopen, int32_t gext) {
    const int64_t L = (int64_t)match * (int64_t)m - T;
    if (L < 0) return -1;
    if (match < 1 || match - mismatch < 1 || gext < 1 || gopen < 1 || L > 2048) return INT32_MAX;
    const int64_t csub = (int64_t)match - mismatch;
    const int64_t c1 = std::min<int64_t>(match, std::min(gopen, gext) + (int64_t)match);
    std::vector<int64_t> fill((size_t)L + 1);              // fill[R]: mismatches and other single units within a loss of R
    for (int64_t R = 0; R <= L; ++R) {
        int64_t b = 0;
        for (int64_t s = 0; s * csub <= R; ++s) b = std::max(b, 2 * s + (R - s * csub) / c1);
        fill[(size_t)R] = b;
    }
    int64_t best = fill[(size_t)L];
    for (int64_t g = 1; g * gopen <= L; ++g)
        for (int64_t len = g; g * gopen + (len - g) * gext <= L; ++len)
            best = std::max(best, len + fill[(size_t)(L - g * gopen - (len - g) * gext)]);
    return best > 1000000 ? INT32_MAX : (int32_t)best;
}

static void fill_dp(DpParams &d, const std::string &ad, int64_t T, const vf_ctx &c, bool is_prefix) {
    d.pat = pack_adapter(ad);
    d.m = (uint32_t)ad.size();
    for (int k = 0; k < 4; ++k) d.eq[k] = 0;
    for (size_t i = 0; i < ad.size(); ++i) d.eq[base_code((unsigned char)ad[i]) & 3u] |= 1u << i;
    d.T = (int32_t)std::max<int64_t>(std::min<int64_t>(T, INT32_MAX), INT32_MIN + 1);
    d.mismatch = c.mismatch;
    d.match = c.match; d.gopen = c.gopen; d.gext = c.gext;
    static const uint32_t coop_max = [] { const char *e = getenv("VFIND_DP_COOP_MAX"); return e ? (uint32_t)strtoul(e, nullptr, 0) : 256u; }();
    d.coop_max = coop_max;
    const int32_t S = 1 << kScoreShift, P = 1 << kPrefShift;
    d.e_open_k = -c.gopen * S + 1;                             // H(pref 0) -> pref 0, len+1
    d.e_ext_k = -c.gext * S + 1 * P + 1;                      // E(pref 0) -> pref 1, len+1
    d.f_open_k = -(c.gopen + c.mismatch) * S + 2 * P;          // H(pref 0) -> pref 2, row bias
    d.f_ext_k = -(c.gext + c.mismatch) * S;                    // F(pref 3) stays 3, row bias
    d.delta = (uint32_t)(c.match - c.mismatch) << kScoreShift;
    d.is_prefix = is_prefix ? 1u : 0u;
    d.independent = 0;
    d.kerr = max_unit_errors(T, ad.size(), c.match, c.mismatch, c.gopen, c.gext);
    // Tightened bound for tasks whose best end column has unit edit distance 1 (nearly all tasks of a
    // low-error library): that alignment loses at most c_max against m matches, so the optimum scores
    // >= match*m - c_max and every optimal alignment has at most kerr_lo unit edits.
    d.kerr_lo = -1;
    if (d.kerr >= 1 && d.kerr != INT32_MAX) {
        const int64_t c_max = std::max<int64_t>(std::max<int64_t>(c.match, (int64_t)c.match - c.mismatch),
                                                (int64_t)c.match + std::max(c.gopen, c.gext));
        const int64_t T_lo = std::max<int64_t>(T, (int64_t)c.match * (int64_t)ad.size() - c_max);
        const int32_t k_lo = max_unit_errors(T_lo, ad.size(), c.match, c.mismatch, c.gopen, c.gext);
        if (k_lo >= 1 && k_lo < d.kerr) d.kerr_lo = k_lo;
    }
    // block filter (k_bfilter): kerr_lo + 1 disjoint stretches of the adapter, the first min(stretch, 8) bases of each
    d.nblk = 0;
    for (int b = 0; b < 4; ++b) { d.blk_off[b] = 0; d.blk_len[b] = 0; }
    if (d.kerr_lo >= 1 && d.kerr_lo <= 3) {
        const uint32_t nb = (uint32_t)d.kerr_lo + 1u, base = d.m / nb, rem = d.m % nb;
        uint32_t o = 0, shortest = 8;
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t ln = base + (b < rem ? 1u : 0u);
            d.blk_off[b] = o; d.blk_len[b] = std::min(ln, 8u);
            shortest = std::min(shortest, d.blk_len[b]);
            o += ln;
        }
        if (shortest >= 5) d.nblk = nb;                 // shorter blocks turn up by chance too often to be worth the pass
    }
    // Hamming certificate: a diagonal on which the adapter differs from the read in h bases is an alignment of score
    // match*m - h*(match - mismatch): the optimum is at least that, and optimal alignments have at most kerr_h[h] unit edits
    for (int h = 0; h < 8; ++h) d.kerr_h[h] = h ? -1 : 0;
    if (d.kerr >= 2 && d.kerr != INT32_MAX) {
        for (int h = 1; h < 8 && h <= d.kerr; ++h) {
            const int64_t T_h = std::max<int64_t>(T, (int64_t)c.match * (int64_t)ad.size() - (int64_t)h * ((int64_t)c.match - c.mismatch));
            const int32_t k_h = max_unit_errors(T_h, ad.size(), c.match, c.mismatch, c.gopen, c.gext);
            if (k_h >= 1 && k_h < d.kerr) { d.kerr_h[h] = k_h; d.kerr_h[0] = 1; }
        }
    }
    // the scan's metric: indel distance where its bound is no weaker than the unit bound (VFIND_INDEL_SCAN=0: never; A/B)
    static const bool indel_ok = [] { const char *e = getenv("VFIND_INDEL_SCAN"); return !(e && !atoi(e)); }();
    d.indel = 0; d.kthr = d.kerr;
    d.dl_lo = d.kerr_lo >= 0 ? d.kerr - d.kerr_lo : 0;
    for (int h = 0; h < 8; ++h) d.dl_h[h] = (h && d.kerr_h[h] >= 0) ? d.kerr - d.kerr_h[h] : 0;
    if (indel_ok && d.kerr >= 1 && d.kerr != INT32_MAX) {
        const int32_t ki = max_indel_units(T, ad.size(), c.match, c.mismatch, c.gopen, c.gext);
        if (ki >= 1 && ki <= d.kerr) {
            d.indel = 1; d.kthr = ki;
            auto shrink = [&](int64_t T_tight) {            // kthr minus the bound under the tightened threshold, in indel units
                const int32_t k = max_indel_units(T_tight, ad.size(), c.match, c.mismatch, c.gopen, c.gext);
                return (k >= 0 && k <= ki) ? ki - k : 0;
            };
            if (d.kerr_lo >= 0) {
                const int64_t c_max = std::max<int64_t>(std::max<int64_t>(c.match, (int64_t)c.match - c.mismatch),
                                                        (int64_t)c.match + std::max(c.gopen, c.gext));
                d.dl_lo = shrink(std::max<int64_t>(T, (int64_t)c.match * (int64_t)ad.size() - c_max));
            }
            for (int h = 1; h < 8; ++h)
                if (d.kerr_h[h] >= 0)
                    d.dl_h[h] = shrink(std::max<int64_t>(T, (int64_t)c.match * (int64_t)ad.size() - (int64_t)h * ((int64_t)c.match - c.mismatch)));
        }
    }
    const uint32_t up = 32u - d.m;                     // d.m in 1..32
    const uint32_t filler = (up && !d.indel) ? ((1u << up) - 1u) : 0u;
    for (int q = 0; q < 4; ++q) d.eq_top[q] = up ? ((d.eq[q] << up) | filler) : d.eq[q];
    d.pv_init = up ? (0xFFFFFFFFu << up) : 0xFFFFFFFFu;
}

static void vf_comm_release(vf_ctx *c);

// ------------------------------------------------------------------------------------------------
// create / destroy
// ------------------------------------------------------------------------------------------------
// all_counters = the table's memory is fresh (creation, the spare's first clear): everything is zeroed.
// Otherwise a letter-key table is cleared by moving to the next epoch (Table::epoch, vf_device.h): its slot lines - 8/9 of
// what a clear wrote - stay as they are and read as free from now on; every 15th clear, when the epoch wraps, zeroes them.
static int table_clear_locked(vf_ctx *c, bool all_counters = false) {
    bool zero_slots = true;
    if (c->epoch_tags) {
        if (all_counters || c->tab.epoch == 0 || c->tab.epoch >= 15) c->tab.epoch = 1;
        else { ++c->tab.epoch; zero_slots = false; }
    } else c->tab.epoch = 0;
    c->tab.key_tag = key_tag_of(c->tab.epoch);
    // arena top, entry count and error bits belong to the table; the read statistics do not
    ClearRegions cr;
    cr.p[0] = c->t_slots.p;    cr.bytes[0] = zero_slots ? c->table_slots * 64 : 0;
    cr.p[1] = c->t_counts.p;   cr.bytes[1] = c->table_slots * sizeof(unsigned long long);
    cr.p[2] = c->t_chunks.p;   cr.bytes[2] = (uint64_t)kChunkRows * kMaxGrid * sizeof(unsigned long long);
    cr.p[3] = c->t_counters.p; cr.bytes[3] = (all_counters ? (uint64_t)C_NUM : (uint64_t)kTableCounters) * sizeof(uint64_t);
    HIP_TRY(launch_clear(c->main_stream, cr));
    // batch streams order their next table access after this clear through the event (no host sync)
    HIP_TRY(hipEventRecord(c->clear_event, c->main_stream));
    return VF_OK;
}
// the table in use <-> the spare (buffers, geometry and the event that says "cleared")
static void table_swap_spare(vf_ctx *c) {
    vf_ctx::Spare &sp = c->spare;
    std::swap(c->t_slots, sp.slots); std::swap(c->t_counts, sp.counts); std::swap(c->t_arena, sp.arena);
    std::swap(c->t_counters, sp.counters); std::swap(c->t_chunks, sp.chunks);
    std::swap(c->tab, sp.tab);
    std::swap(c->table_slots, sp.table_slots); std::swap(c->arena_dwords, sp.arena_dwords);
    std::swap(c->clear_event, sp.cleared);
}
// first clear of a context: a second table of the first one's present size, cleared (on the main stream)
static int table_make_spare(vf_ctx *c) {
    vf_ctx::Spare &sp = c->spare;
    int rc;
    if ((rc = sp.slots.ensure(c->table_slots * 4)) || (rc = sp.counts.ensure(c->table_slots)) ||
        (rc = sp.arena.ensure(c->arena_dwords)) || (rc = sp.counters.ensure(C_NUM)) ||
        (rc = sp.chunks.ensure((size_t)kChunkRows * kMaxGrid))) {
        sp.slots.release(); sp.counts.release(); sp.arena.release(); sp.counters.release(); sp.chunks.release();
        return rc;
    }
    if (!sp.cleared) HIP_TRY(hipEventCreateWithFlags(&sp.cleared, hipEventDisableTiming));
    sp.tab = c->tab;
    sp.tab.chunk_state = sp.chunks.p; sp.tab.slots = sp.slots.p; sp.tab.counts = sp.counts.p;
    sp.tab.arena = sp.arena.p; sp.tab.counters = sp.counters.p;
    sp.table_slots = c->table_slots; sp.arena_dwords = c->arena_dwords;
    table_swap_spare(c);                                   // clear it through the one function that knows the regions
    rc = table_clear_locked(c, true);
    table_swap_spare(c);
    if (rc) return rc;
    sp.valid = true;
    return VF_OK;
}

static int vf_ctx_create_impl(const vf_params *p, vf_ctx **out);
extern "C" int vf_ctx_create(const vf_params *p, vf_ctx **out) { return vf_guard([&] { return vf_ctx_create_impl(p, out); }); }
static int vf_ctx_create_impl(const vf_params *p, vf_ctx **out) {
    if (!p || !out) return fail(VF_EINVAL, "null argument");
    *out = nullptr;
    if ((p->prefix_len && !p->prefix) || (p->suffix_len && !p->suffix)) return fail(VF_EINVAL, "null adapter");
    // every early return below destroys what has been created so far (streams, events, device buffers)
    struct CtxDeleter { void operator()(vf_ctx *p) const { vf_ctx_destroy(p); } };
    std::unique_ptr<vf_ctx, CtxDeleter> c(new vf_ctx);
    // prefix threshold is validated first (lib.rs:239), then suffix (lib.rs:249)
    int rc;
    if ((rc = preflight(p->accept_prefix_alignment, &c->skip_pre))) return rc;
    if ((rc = preflight(p->accept_suffix_alignment, &c->skip_suf))) return rc;
    c->pre.assign((const char *)p->prefix, p->prefix_len);
    c->suf.assign((const char *)p->suffix, p->suffix_len);
    c->match = p->match_score; c->mismatch = p->mismatch_score;
    c->gopen = p->gap_open_penalty; c->gext = p->gap_extend_penalty;
    c->T_pre = int_threshold(p->accept_prefix_alignment, p->match_score, c->pre.size());
    c->T_suf = int_threshold(p->accept_suffix_alignment, p->match_score, c->suf.size());
    c->skip_translation = p->skip_translation != 0;
    // every key of a translating context is made of letters (ASCII): room for the epoch in each byte's top bit
    { const char *e = getenv("VFIND_EPOCH_TAGS"); c->epoch_tags = !c->skip_translation && !(e && atoi(e) == 0); }
    c->n_threads = p->n_threads ? p->n_threads : 3;
    c->queue_len = std::min<uint32_t>(std::max<uint32_t>(p->queue_len, 1), 16);
    c->batch_reads = p->batch_reads ? p->batch_reads : (1u << 18);
    c->dev = p->device;
    {
        const uint64_t want = p->table_slots ? p->table_slots : (1ull << 24);
        uint64_t pow2 = 1; while (pow2 < want && pow2 < (1ull << 33)) pow2 <<= 1;
        if (pow2 > (1ull << 32)) return fail(VF_EINVAL, "table_slots too large");
        c->table_slots = pow2;
    }

    // can the 2-bit fast path hold these adapters and scores?  (else every read goes byte-wise)
    auto small = [](int32_t v) { return v >= -127 && v <= 127; };
    c->fast_ok = !c->pre.empty() && !c->suf.empty() && c->pre.size() <= (size_t)kMaxFastAdapter &&
                 c->suf.size() <= (size_t)kMaxFastAdapter && pure_acgt(c->pre) && pure_acgt(c->suf) && small(c->match) &&
                 small(c->mismatch) && c->gopen >= 0 && c->gopen <= 127 && c->gext >= 0 && c->gext <= 127 &&
                 c->match >= c->mismatch;

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(VF_EDEVICE, "no HIP device visible: the HIP path is required (no CPU fallback)");
    if (c->dev < 0 || c->dev >= ndev) return fail(VF_EINVAL, "device ordinal out of range");
    HIP_TRY(hipSetDevice(c->dev));
    HIP_TRY(hipStreamCreateWithFlags(&c->main_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->clear_event, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->fin_event, hipEventDisableTiming));
    if (const char *e = getenv("VFIND_BLOCK_FILTER")) c->block_filter = atoi(e) != 0;
    if (const char *e = getenv("VFIND_TABLE_SPARE")) c->spare_on = atoi(e) != 0;
    if (const char *e = getenv("VFIND_RES_LANES")) c->n_res = (uint32_t)std::min(std::max(atoi(e), 1), (int)vf_ctx::kMaxResLanes);
    for (uint32_t i = 0; i < c->n_res; ++i) {
        ResLane &L = c->res[i];
        HIP_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
        L.chunk_row = 17 + i;
        if ((rc = L.h_entries.ensure(1))) return rc;
        *L.h_entries.p = 0;
    }

    // adapters on the device (byte-wise kernel)
    const size_t alen = c->pre.size() + c->suf.size();
    if ((rc = c->d_adapters.ensure(alen + 16))) return rc;
    if (!c->pre.empty()) HIP_TRY(hipMemcpy(c->d_adapters.p, c->pre.data(), c->pre.size(), hipMemcpyHostToDevice));
    if (!c->suf.empty()) HIP_TRY(hipMemcpy(c->d_adapters.p + c->pre.size(), c->suf.data(), c->suf.size(), hipMemcpyHostToDevice));
    c->gp.pre = c->d_adapters.p; c->gp.suf = c->d_adapters.p + c->pre.size();
    c->gp.pre_len = (uint32_t)c->pre.size(); c->gp.suf_len = (uint32_t)c->suf.size();
    c->gp.align_pre = c->skip_pre ? 0 : 1; c->gp.align_suf = c->skip_suf ? 0 : 1;
    c->gp.T_pre = c->T_pre; c->gp.T_suf = c->T_suf;
    c->gp.match = c->match; c->gp.mismatch = c->mismatch; c->gp.open = c->gopen; c->gp.ext = c->gext;
    c->gp.skip_translation = c->skip_translation ? 1 : 0;
    c->gp.count_mode = 1;
    if (c->fast_ok) {
        c->sp.pre_pat = pack_adapter(c->pre); c->sp.suf_pat = pack_adapter(c->suf);
        c->sp.pre_len = (uint32_t)c->pre.size(); c->sp.suf_len = (uint32_t)c->suf.size();
        c->sp.align_pre = c->skip_pre ? 0 : 1; c->sp.align_suf = c->skip_suf ? 0 : 1;
        fill_dp(c->dpP, c->pre, c->T_pre, *c, true);
        fill_dp(c->dpS, c->suf, c->T_suf, *c, false);
        // window = m + kerr columns (+ spread of the end columns) must fit the 64 the kernel holds
        auto usable = [](const DpParams &d) { return d.kerr < 0 || (int64_t)d.m + d.kerr <= 48; };
        c->filter_on = usable(c->dpP) && usable(c->dpS);
    }

    // count table
    c->arena_dwords = (p->arena_bytes ? p->arena_bytes : (1ull << 30)) / 4;
    if ((rc = c->t_slots.ensure(c->table_slots * 4))) return rc;
    if ((rc = c->t_counts.ensure(c->table_slots))) return rc;
    if ((rc = c->t_arena.ensure(c->arena_dwords))) return rc;
    if ((rc = c->t_counters.ensure(C_NUM))) return rc;
    if ((rc = c->t_chunks.ensure((size_t)kChunkRows * kMaxGrid))) return rc;
    c->tab.chunk_state = c->t_chunks.p; c->tab.chunk_row = 16;      // the main stream (merge)
    c->tab.slots = c->t_slots.p;
    c->tab.counts = c->t_counts.p;
    c->tab.arena = c->t_arena.p; c->tab.counters = c->t_counters.p;
    c->tab.slot_mask = (uint32_t)(c->table_slots - 1); c->tab.arena_cap = c->arena_dwords;
    if ((rc = table_clear_locked(c.get(), true))) return rc;

    for (uint32_t i = 0; i < c->queue_len; ++i) {
        c->slots.emplace_back(new Slot);                    // owned by the context first: a failure below frees it with the rest
        Slot *s = c->slots.back().get();
        s->index = i;
        HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
        if ((rc = s->h_entries.ensure(1))) return rc;
        *s->h_entries.p = 0;
    }
    *out = c.release();
    return VF_OK;
}

static void vf_ctx_destroy_impl(vf_ctx *c);
extern "C" void vf_ctx_destroy(vf_ctx *c) {
    try { vf_ctx_destroy_impl(c); } catch (...) { }
}
static void vf_ctx_destroy_impl(vf_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->dev);
    (void)hipDeviceSynchronize();
    for (auto &e : c->pending_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &s : c->slots) {
        s->h_rows.release(); s->h_lens.release(); s->d_rows.release(); s->d_lens.release(); s->ws.release();
        s->h_gbytes.release(); s->h_goffs.release(); s->d_gbytes.release(); s->d_goffs.release();
        s->d_gstart.release(); s->d_gend.release(); s->d_gws.release(); s->h_gstart.release(); s->h_gend.release();
        s->h_spans.release();
        s->h_text.release(); s->d_text.release(); s->d_tile_cnt.release(); s->d_tile_off.release(); s->d_line_end.release();
        s->d_info.release(); s->d_gen_start.release(); s->d_gen_len.release(); s->h_info.release();
        s->h_entries.release();
        if (s->done) (void)hipEventDestroy(s->done);
        if (s->stream) (void)hipStreamDestroy(s->stream);
    }
    for (int i = 0; i < vf_ctx::kMaxResLanes; ++i) {
        ResLane &L = c->res[i];
        L.ws.release(); L.h_entries.release();
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    c->exp_refs.release(); c->exp_counts.release(); c->d_part.release(); c->d_comm_words.release(); c->h_hdr.release();
    vf_comm_release(c);
    c->t_slots.release(); c->t_counts.release(); c->t_arena.release(); c->t_counters.release(); c->t_chunks.release();
    c->spare.slots.release(); c->spare.counts.release(); c->spare.arena.release(); c->spare.counters.release(); c->spare.chunks.release();
    if (c->spare.cleared) (void)hipEventDestroy(c->spare.cleared);
    c->d_adapters.release();
    if (c->clear_event) (void)hipEventDestroy(c->clear_event);
    if (c->fin_event) (void)hipEventDestroy(c->fin_event);
    if (c->main_stream) (void)hipStreamDestroy(c->main_stream);
    delete c;
}

// ------------------------------------------------------------------------------------------------
// profiling events
// ------------------------------------------------------------------------------------------------
static int resolve_events(vf_ctx *c) {
    std::lock_guard<std::mutex> g(c->ev_mu);
    for (auto &e : c->pending_events) {
        float ms = 0.f;
        hipError_t r = hipEventSynchronize(e.b);
        if (r == hipSuccess) r = hipEventElapsedTime(&ms, e.a, e.b);
        if (r == hipSuccess) { c->ms[e.kind] += ms; c->launches[e.kind]++; }
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    c->pending_events.clear();
    return VF_OK;
}
struct Timed {
    vf_ctx *c; hipStream_t s; int kind; hipEvent_t a = nullptr, b = nullptr; bool on;
    // mode = the profiling mode as the submitting call read it ONCE, sampled = that call's batch is one of mode 3's sampled
    // ones: both decided per call (several threads may submit; a shared flag switched one batch's brackets from another's)
    Timed(vf_ctx *c_, hipStream_t s_, int k, int mode, bool sampled) : c(c_), s(s_), kind(k), on(mode == 1 || ((mode == 2 || (mode == 3 && sampled)) && (k == K_FIN || k == K_SCAN))) {
        if (on) {
            // timing only: nothing synchronizes-with these events, so they carry no system-scope fence (hip_runtime_api.h recommends it
            // for events that only measure: a default record writes the caches back; EXPERIMENTS R4.10: no difference measured here)
            if (hipEventCreateWithFlags(&a, hipEventDisableSystemFence) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableSystemFence) != hipSuccess) { on = false; return; }
            (void)hipEventRecord(a, s);
        }
    }
    ~Timed() {
        if (on) {
            (void)hipEventRecord(b, s);
            std::lock_guard<std::mutex> g(c->ev_mu);
            c->pending_events.push_back({a, b, kind});
        }
    }
};

// ------------------------------------------------------------------------------------------------
// table growth: when more than half the slots hold keys, quadruple the table and re-insert on the device
// ------------------------------------------------------------------------------------------------
static int sync_all_streams(vf_ctx *c) {
    for (auto &s : c->slots) HIP_TRY(hipStreamSynchronize(s->stream));
    for (auto &L : c->res) if (L.stream) HIP_TRY(hipStreamSynchronize(L.stream));
    HIP_TRY(hipStreamSynchronize(c->main_stream));
    return VF_OK;
}

// every (stream row, workgroup) that has not taken an arena chunk yet may take one for a single key
static inline uint64_t arena_slack(const vf_ctx *c) { return (uint64_t)(c->queue_len + vf_ctx::kMaxResLanes + 1) * kMaxGrid * kChunk; }

// grows until (entries + extra) fits in half the slots and the arena has room for extra_arena more dwords; extra =
// reads about to be submitted, or a peer's entries about to be merged (each may be a new key).  Re-inserted LONG keys
// keep their arena offsets (k_rehash), so growth itself takes no arena room.  Takes the table lock exclusively: no
// submitter is enqueueing while the table is swapped.
static int table_grow_locked(vf_ctx *c, bool use_unenqueued, uint64_t extra, uint64_t extra_arena = 0) {
    std::unique_lock<std::shared_mutex> lk(c->table_mu);
    int rc;
    if ((rc = sync_all_streams(c))) return rc;
    uint64_t h[C_NUM];
    HIP_TRY(hipMemcpy(h, c->t_counters.p, sizeof(h), hipMemcpyDeviceToHost));
    if (use_unenqueued) {
        // a streaming submitter ran out of (conservatively counted) room: make room for what is on its way in AND for a
        // full pipeline behind it - every slot's latest batch - so that the bound capacity_refresh can give without
        // draining the streams fits from now on.  Sizing for the reads on their way in alone left a table that was a tenth
        // full and a drain of every stream every two or three batches (memory is not what an MI355X is short of).
        uint64_t pipeline = 0;
        for (auto &s : c->slots) pipeline += s->inflight_reads.load();
        std::lock_guard<std::mutex> g(c->cap_mu);
        extra = c->unenqueued + pipeline; extra_arena = c->unenqueued_arena;
    }
    // key arena: offsets stay valid, so growing is a copy
    if (extra_arena) {
        uint64_t cap = c->arena_dwords;
        while (h[C_ARENA_TOP] + extra_arena + arena_slack(c) > cap) cap *= 4;
        if (cap != c->arena_dwords) {
            if (cap > (1ull << 36)) return fail(VF_ECAPACITY, "device key arena cannot grow beyond 256 GiB");
            DevBuf<uint32_t> n_arena;
            if ((rc = n_arena.ensure(cap))) return rc;
            HIP_TRY(hipMemcpy(n_arena.p, c->t_arena.p, std::min<uint64_t>(h[C_ARENA_TOP], c->arena_dwords) * 4, hipMemcpyDeviceToDevice));
            c->t_arena.release();
            c->t_arena = n_arena;
            c->arena_dwords = cap;
            c->tab.arena = c->t_arena.p; c->tab.arena_cap = cap;
        }
    }
    while ((h[C_ENTRIES] + extra) * 2 > c->table_slots) {
        const uint64_t new_slots = c->table_slots * 4;
        if (new_slots > (1ull << 32)) return fail(VF_ECAPACITY, "device count table cannot grow beyond 2^32 slots");
        DevBuf<uint4> n_slots; DevBuf<unsigned long long> n_counts;
        if ((rc = n_slots.ensure(new_slots * 4))) return rc;
        if ((rc = n_counts.ensure(new_slots))) { n_slots.release(); return rc; }
        HIP_TRY(hipMemsetAsync(n_slots.p, 0, new_slots * 64, c->main_stream));
        HIP_TRY(hipMemsetAsync(n_counts.p, 0, new_slots * sizeof(unsigned long long), c->main_stream));
        HIP_TRY(hipMemsetAsync(c->t_counters.p + C_ENTRIES, 0, sizeof(uint64_t), c->main_stream));     // recounted by the re-insert
        HIP_TRY(hipMemsetAsync(c->t_counters.p + C_KEY_DWORDS, 0, sizeof(uint64_t), c->main_stream));
        Table to = c->tab;
        to.slots = n_slots.p; to.counts = n_counts.p; to.slot_mask = (uint32_t)(new_slots - 1);
        HIP_TRY(launch_rehash(c->main_stream, c->tab, to));
        HIP_TRY(hipStreamSynchronize(c->main_stream));
        c->t_slots.release(); c->t_counts.release();
        c->t_slots = n_slots; c->t_counts = n_counts;
        c->table_slots = new_slots;
        c->tab = to;
        HIP_TRY(hipMemcpy(h, c->t_counters.p, sizeof(h), hipMemcpyDeviceToHost));
    }
    // (every stream has been drained: nothing is in flight, the bound is exact again)
    {
        std::lock_guard<std::mutex> g(c->cap_mu);
        c->upper_entries = h[C_ENTRIES] + c->unenqueued; c->upper_arena = h[C_ARENA_TOP] + c->unenqueued_arena;
        c->known_entries = h[C_ENTRIES]; c->pending_reads = 0;
        for (auto &s : c->slots) s->cap_pending = false;
    }
    for (auto &s : c->slots) if (s->h_entries.p) *s->h_entries.p = h[C_ENTRIES];
    for (auto &L : c->res) if (L.h_entries.p) *L.h_entries.p = h[C_ENTRIES];
    return VF_OK;
}

// resident path: the previous batch brought the entry count back; grow when more than half is used
static int table_grow_if_needed(vf_ctx *c, uint64_t entries_hint) {
    if (entries_hint * 2 <= c->table_slots) return VF_OK;
    return table_grow_locked(c, false, 0);
}

// streaming paths: every read of a batch may add a key, so a batch is only enqueued when the table
// could take all of them (a conservative count, refreshed from the device whenever it runs out)
struct CapacityTicket {
    vf_ctx *c; uint64_t n = 0, arena = 0;
    ~CapacityTicket() { if (n) { std::lock_guard<std::mutex> g(c->cap_mu); c->unenqueued -= n; c->unenqueued_arena -= arena; } }
};
// worst-case arena dwords one read can take: only keys longer than the 14 inline dwords go there
// (56 letters = 168 bases translated, 224 bases untranslated); a workgroup wastes < 1/4 chunk per refill,
// keys beyond a quarter chunk take whole chunks
static inline uint64_t arena_need(const vf_ctx *c, uint32_t len) {
    if (len <= (c->skip_translation ? 224u : 168u)) return 0;
    const uint64_t nd = c->skip_translation ? (len + 15) / 16 : (len / 3 + 3) / 4;
    return nd > kChunk / 4 ? nd + kChunk : nd + nd / 3 + 1;
}
// A batch has been enqueued on a slot: its reads count as possible new keys until the batch is seen to have finished.
static void capacity_note_enqueued(vf_ctx *c, Slot *s, uint64_t n) {
    {
        std::lock_guard<std::mutex> g(c->cap_mu);
        s->inflight_reads = n;
        c->pending_reads += n;
        s->cap_seq.fetch_add(1);
        s->cap_pending = true;
    }
    s->in_flight = true;
}
// ... and has finished (its `done` event is complete): the entry count it brought back stands for its reads from now on.
// seq = the batch the caller saw finish (kAnySeq: the slot's owner, who has waited for the event itself).  Another thread's
// refresh may have queried `done` just before the owner re-recorded it for the NEXT batch: without the sequence check it
// would take the new batch's reads off the books and trust an entry count from before that batch.
constexpr uint64_t kAnySeq = ~0ull;
static void capacity_retire(vf_ctx *c, Slot *s, uint64_t seq = kAnySeq) {
    std::lock_guard<std::mutex> g(c->cap_mu);
    if (!s->cap_pending.load()) return;
    if (seq != kAnySeq && s->cap_seq.load() != seq) return;
    s->cap_pending = false;
    const uint64_t n = s->inflight_reads.load();
    c->pending_reads -= std::min(c->pending_reads, n);
    if (s->h_entries.p && *s->h_entries.p > c->known_entries) c->known_entries = *s->h_entries.p;
}
// The conservative entry count brought up to date WITHOUT draining the streams and without any device work: every batch
// whose event has completed is retired (no wait: hipEventQuery), and the bound is the largest entry count a finished batch
// brought back + the reads still in flight + the reads on their way in.  The running bound otherwise grows by a batch with
// every submit and sent the submitter into table_grow_locked - a drain of every stream - every two or three batches
// although the table was a tenth full: the copies of the PCIe-inclusive leg ran in pairs with 0.6 ms of idle link behind
// each pair.  (Asking the device instead - a tiny kernel or copy on a stream of its own - waits behind whichever slot
// shares its hardware queue, and that slot's kernels wait for its 47 MB copy: 5 ms.)
static bool capacity_refresh(vf_ctx *c, uint64_t *upper_out) {
    // resident batches (vf_process_device_packed) are not in these books while they run; once idle, the entry count their
    // lane brought back is as good as a slot's (a context may mix the resident and the streaming API)
    uint64_t res_known = 0;
    for (auto &L : c->res) {
        if (!L.stream) continue;
        if (L.in_flight.load() && hipEventQuery(L.done) != hipSuccess) return false;
        if (L.h_entries.p) res_known = std::max<uint64_t>(res_known, *L.h_entries.p);
    }
    for (auto &s : c->slots) {
        const uint64_t seq = s->cap_seq.load();                        // before the query: a batch noted later has a later number
        if (s->cap_pending.load() && s->in_flight.load() && hipEventQuery(s->done) == hipSuccess) capacity_retire(c, s.get(), seq);
    }
    std::lock_guard<std::mutex> g(c->cap_mu);
    if (res_known > c->known_entries) c->known_entries = res_known;
    const uint64_t fresh = c->known_entries + c->pending_reads + c->unenqueued;
    if (fresh < c->upper_entries) c->upper_entries = fresh;
    *upper_out = c->upper_entries;
    return true;
}
static int reserve_capacity(vf_ctx *c, CapacityTicket &t, uint64_t n, uint64_t arena) {
    uint64_t up, upa;
    {
        std::lock_guard<std::mutex> g(c->cap_mu);
        t.n = n; t.arena = arena;
        c->unenqueued += n; c->unenqueued_arena += arena;
        up = (c->upper_entries += n); upa = (c->upper_arena += arena);
    }
    const bool arena_ok = !arena || upa + arena_slack(c) <= c->arena_dwords;
    if (up * 2 <= c->table_slots && arena_ok) return VF_OK;
    if (arena_ok && capacity_refresh(c, &up) && up * 2 <= c->table_slots) return VF_OK;
    return table_grow_locked(c, true, 0);
}

// ------------------------------------------------------------------------------------------------
// device path over one packed batch (rows/lens already in HBM)
// ------------------------------------------------------------------------------------------------
static int run_packed(vf_ctx *c, hipStream_t st, uint32_t chunk_row, Workspace &ws, const uint32_t *d_rows, int W,
                      const uint16_t *d_lens, uint64_t n, short2 *d_spans, bool count, uint64_t *h_entries) {
    if (n == 0) return VF_OK;
    if (count) c->part_world = 0;
    if (n > 0x80000000ull) return fail(VF_EINVAL, "batch too large (max 2^31 reads per submission)");
    if (count) { const int lrc = ws.ensure_log(n, c->main_stream); if (lrc) return lrc; }
    // every third batch: an odd period, so that the sampled batches alternate between the two batch lanes (every fourth sat on
    // one lane, always in the same phase against the other lane's k_finish: k_scan 0.515 ms against 0.43-0.47 over all launches)
    const int pmode = c->profiling.load();
    static const uint32_t prof_period = [] { const char *e = getenv("VFIND_PROF_PERIOD"); const long v = e ? atol(e) : 3; return (uint32_t)(v < 1 ? 1 : v); }();
    const bool psample = pmode == 3 && c->prof_batch.fetch_add(1) % prof_period == 0u;
    BatchLists bl{};
    bl.rows = d_rows; bl.lens = d_lens;
    bl.listP = ws.listP.p; bl.listS = ws.listS.p; bl.listF = ws.listF.p;
    bl.cntP = ws.cnt.p; bl.cntS = ws.cnt.p + kMaxGrid; bl.cntF = ws.cnt.p + 2 * kMaxGrid;
    bl.winW2 = ws.winW2.p;
    bl.listN = ws.listN.p; bl.winN = ws.winN.p; bl.listR = ws.listR.p;
    bl.cntN = ws.cnt.p + 7 * kMaxGrid; bl.cntR = ws.cnt.p + 8 * kMaxGrid; bl.cntB = ws.cnt.p + 9 * kMaxGrid;
    bl.listB = ws.listB.p; bl.winB = ws.winB.p;
    bl.tot = ws.cnt.p + 10 * kMaxGrid;
    for (int q = 0; q < 4; ++q) { bl.listW[q] = ws.listW[q].p; bl.winW[q] = ws.winW[q].p; bl.cntW[q] = ws.cnt.p + (3 + q) * kMaxGrid; }
    scan_geometry(n, W, &bl.grid, &bl.seg);
    // counting mode skips the suffix alignment of a read whose prefix failed (same table, less work);
    // the spans test hook keeps the two searches independent, as find_adapter_match is called (lib.rs:278-286)
    ScanParams sp = c->sp; DpParams dpP = c->dpP, dpS = c->dpS;
    sp.independent = dpP.independent = dpS.independent = count ? 0u : 1u;
    { Timed t(c, st, K_SCAN, pmode, psample); HIP_TRY(launch_scan(st, W, d_rows, d_lens, n, sp, d_spans, bl)); }
    // per adapter: filter (block filter + Myers pass over what it leaves, one launch; the Myers pass alone where the scoring
    // gives the block filter no bound) -> every task class of the DP in one launch; or the full-width DP alone when the
    // scoring scheme gives no edit bound at all
    for (int role = 0; role < 2; ++role) {
        if (role == 0 ? c->skip_pre : c->skip_suf) continue;
        const DpParams &dp = role == 0 ? dpP : dpS;
        if (c->filter_on) {
            {
                Timed t(c, st, K_FILTER, pmode, psample);
                if (dp.nblk != 0 && c->block_filter) HIP_TRY(launch_bmfilter(st, W, dp, d_spans, bl, c->tab.counters));
                else HIP_TRY(launch_filter(st, W, dp, d_spans, bl, c->tab.counters));
            }
            { Timed t(c, st, K_DPW, pmode, psample); HIP_TRY(launch_dp_classes(st, W, dp, d_spans, bl, c->tab.counters)); }
        } else {
            Timed t(c, st, K_DPREST, pmode, psample);
            HIP_TRY(launch_dp(st, W, dp, d_spans, bl, c->tab.counters));
        }
    }
    Table tab = c->tab;
    tab.chunk_row = chunk_row;
    // only the table kernels are ordered after the latest clear; scan and DP may overlap it
    if (count) HIP_TRY(hipStreamWaitEvent(st, c->clear_event, 0));
    // the previous log of this workspace has been applied (long ago, as a rule)
    if (count && ws.log_pending) { HIP_TRY(hipStreamWaitEvent(st, ws.log_applied, 0)); ws.log_pending = false; }
    if (count) {
        // With a spare table nothing else orders the table kernels of two batch lanes: side by side they take 40 % longer
        // each.  Wait, launch and record under one lock: several submitting threads then chain their table kernels one
        // behind the other instead of all waiting for the same earlier record.
        std::lock_guard<std::mutex> fg(c->fin_mu);
        if (c->spare.valid && c->fin_recorded.load()) HIP_TRY(hipStreamWaitEvent(st, c->fin_event, 0));
        { Timed t(c, st, K_FIN, pmode, psample); HIP_TRY(launch_finish(st, W, d_rows, d_lens, n, d_spans, c->skip_translation, tab, ws.count_log())); }
        if (c->spare.valid) { HIP_TRY(hipEventRecord(c->fin_event, st)); c->fin_recorded.store(true); }
    }
    if (count) {
        // The log is applied on the main stream: the batch's own stream goes on with its next batch, the other lane's
        // k_finish (which reads no count) starts beside it, and whatever reads or drops counts - clear, export, merge,
        // growth - is on the main stream behind it or synchronises it.
        HIP_TRY(hipEventRecord(ws.log_written, st));
        std::lock_guard<std::mutex> g(c->log_mu);                       // one log's kernels at a time on the main stream
        HIP_TRY(hipStreamWaitEvent(c->main_stream, ws.log_written, 0));
        {
            Timed t(c, c->main_stream, K_LOG, pmode, psample);
            HIP_TRY(launch_count_log(c->main_stream, W, c->skip_translation, n, tab, ws.count_log()));
            const CountLog l = ws.count_log();                          // part totals and scatter cursors: zero for the next batch
            HIP_TRY(hipMemsetAsync(l.tot, 0, 4 * (size_t)kLogMaxParts * sizeof(uint32_t), c->main_stream));
        }
        HIP_TRY(hipEventRecord(ws.log_applied, c->main_stream));
        ws.log_pending = true;
    }
    // the entry count rides back with the batch: the host grows the table between batches when it fills
    if (count && h_entries) HIP_TRY(launch_peek(st, c->t_counters.p + C_ENTRIES, h_entries, 2));
    return VF_OK;
}

// generic (byte-wise) batch already staged in slot pinned buffers
static int run_generic(vf_ctx *c, Slot &s, uint64_t ng, uint64_t nbytes, uint32_t max_len, bool count, bool want_spans) {
    if (ng == 0) return VF_OK;
    int rc;
    if ((rc = s.d_gbytes.ensure(nbytes + 16))) return rc;
    if ((rc = s.d_goffs.ensure(ng + 1))) return rc;
    HIP_TRY(hipMemcpyAsync(s.d_gbytes.p, s.h_gbytes.p, nbytes, hipMemcpyHostToDevice, s.stream));
    HIP_TRY(hipMemcpyAsync(s.d_goffs.p, s.h_goffs.p, (ng + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s.stream));
    const uint32_t mmax = (uint32_t)std::max(c->pre.size(), c->suf.size());
    const uint32_t rows = 4 * mmax + (max_len + 3) / 4 + 4;
    uint64_t threads = std::min<uint64_t>((ng + 255) / 256 * 256, 65536);
    if ((rc = s.d_gws.ensure((size_t)rows * threads))) return rc;
    int32_t *ds = nullptr, *de = nullptr;
    if (want_spans) {
        if ((rc = s.d_gstart.ensure(ng))) return rc;
        if ((rc = s.d_gend.ensure(ng))) return rc;
        ds = s.d_gstart.p; de = s.d_gend.p;
    }
    GenericParams gp = c->gp;
    gp.count_mode = count ? 1 : 0;
    Table tab = c->tab;
    tab.chunk_row = s.index;
    if (count) HIP_TRY(hipStreamWaitEvent(s.stream, c->clear_event, 0));
    { Timed t(c, s.stream, K_GEN, c->profiling.load(), false); HIP_TRY(launch_generic(s.stream, s.d_gbytes.p, s.d_goffs.p, ng, gp, ds, de, s.d_gws.p, threads, rows, tab)); }
    if (count && s.h_entries.p) HIP_TRY(launch_peek(s.stream, c->t_counters.p + C_ENTRIES, s.h_entries.p, 2));
    return VF_OK;
}

// ------------------------------------------------------------------------------------------------
// slots
// ------------------------------------------------------------------------------------------------
// The free slot that was handed out longest ago: a single submitting thread then walks round the slots, so the copy of
// batch k + 1 really overlaps the kernels of batch k (taking the FIRST free slot gave that thread slot 0 every time and it
// waited for its own previous batch: no overlap at all unless several threads submitted at once).
static Slot *acquire_slot(vf_ctx *c) {
    std::unique_lock<std::mutex> lk(c->mu);
    Slot *s = nullptr;
    c->cv.wait(lk, [&] {
        for (auto &x : c->slots) if (!x->busy && (!s || x->last_use < s->last_use)) s = x.get();
        return s != nullptr;
    });
    s->busy = true;
    s->last_use = ++c->slot_ticket;
    return s;
}
static void release_slot(vf_ctx *c, Slot *s) {
    { std::lock_guard<std::mutex> g(c->mu); s->busy = false; }
    c->cv.notify_one();
}
static const char *fastq_error_text(uint32_t code) {
    switch (code) {
    case 1: return "malformed FASTQ: blank line inside the file";
    case 2: return "malformed FASTQ: record does not start with '@'";
    case 3: return "malformed FASTQ: missing '+' separator line";
    case 4: return "malformed FASTQ: sequence and quality lengths differ";
    default: return "malformed FASTQ";
    }
}
// outcome of the slot's latest device-framed batch (its stream must have been waited for)
static int collect_text_result(vf_ctx *c, Slot *s) {
    if (!s->text_pending) return VF_OK;
    s->text_pending = false;
    c->n_generic += s->h_info.p[3];
    const uint32_t e = s->h_info.p[2];
    if (e != 0xFFFFFFFFu) {
        vf_internal_set_async_error(c, VF_EFORMAT, fastq_error_text(e & 0xFFu));
        return fail(VF_EFORMAT, fastq_error_text(e & 0xFFu));
    }
    return VF_OK;
}
static int wait_slot(vf_ctx *c, Slot *s) {
    if (s->in_flight) {
        HIP_TRY(hipEventSynchronize(s->done));
        capacity_retire(c, s);
        s->in_flight = false;
    }
    return collect_text_result(c, s);
}

// classify + pack one chunk into a slot and enqueue it.  spans_out != nullptr => synchronous,
// spans-only mode (vf_spans_ascii).
static int submit_chunk(vf_ctx *c, const uint8_t *base, const ReadRef *reads, uint64_t n, int32_t *starts_out, int32_t *ends_out) {
    if (n == 0) return VF_OK;
    HIP_TRY(hipSetDevice(c->dev));
    const bool spans_mode = starts_out != nullptr;
    Slot *s = acquire_slot(c);
    struct Rel { vf_ctx *c; Slot *s; ~Rel() { release_slot(c, s); } } rel{c, s};
    int rc;
    if ((rc = wait_slot(c, s))) return rc;
    CapacityTicket ticket{c};
    if (!spans_mode) {
        uint64_t need = 0;
        for (uint64_t i = 0; i < n; ++i) need += arena_need(c, reads[i].len);
        if ((rc = reserve_capacity(c, ticket, n, need))) return rc;
    }
    std::shared_lock<std::shared_mutex> table_lock(c->table_mu);

    // geometry of the fast part
    uint32_t max_fast = 0;
    if (c->fast_ok)
        for (uint64_t i = 0; i < n; ++i) if (reads[i].len <= (uint32_t)kMaxFastRead) max_fast = std::max(max_fast, reads[i].len);
    const uint32_t W = c->fast_ok ? std::max(vf_row_dwords_for(max_fast), 10u) : 0;
    if (W) {
        if ((rc = s->h_rows.ensure((size_t)n * W))) return rc;
        if ((rc = s->h_lens.ensure(n))) return rc;
    }
    s->fast_idx.clear(); s->gen_idx.clear();
    uint64_t nf = 0, ng = 0, gbytes = 0;
    uint32_t gmax = 0;
    // generic staging is sized for the worst case lazily: first pass packs, failures are appended
    std::vector<uint64_t> gen_list;
    for (uint64_t i = 0; i < n; ++i) {
        const ReadRef &r = reads[i];
        bool fast = W && r.len <= (uint32_t)kMaxFastRead;
        if (fast) fast = vf_pack_read(base + r.off, r.len, s->h_rows.p + nf * W, W) == 0;
        if (fast) {
            s->h_lens.p[nf] = (uint16_t)r.len;
            if (spans_mode) s->fast_idx.push_back((uint32_t)i);
            ++nf;
        } else {
            gen_list.push_back(i);
            gbytes += r.len;
            gmax = std::max(gmax, r.len);
        }
    }
    ng = gen_list.size();
    if (ng) {
        if ((rc = s->h_gbytes.ensure(gbytes + 16))) return rc;
        if ((rc = s->h_goffs.ensure(ng + 1))) return rc;
        uint64_t top = 0;
        for (uint64_t k = 0; k < ng; ++k) {
            const ReadRef &r = reads[gen_list[k]];
            memcpy(s->h_gbytes.p + top, base + r.off, r.len);
            s->h_goffs.p[k] = top;
            top += r.len;
            if (spans_mode) s->gen_idx.push_back((uint32_t)gen_list[k]);
        }
        s->h_goffs.p[ng] = top;
    }
    c->n_reads += n;
    c->n_generic += ng;

    if (nf) {
        if ((rc = s->d_rows.ensure((size_t)nf * W))) return rc;
        if ((rc = s->d_lens.ensure(nf))) return rc;
        if ((rc = s->ws.ensure(nf, W))) return rc;
        HIP_TRY(hipMemcpyAsync(s->d_rows.p, s->h_rows.p, (size_t)nf * W * 4, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(s->d_lens.p, s->h_lens.p, nf * sizeof(uint16_t), hipMemcpyHostToDevice, s->stream));
        if ((rc = run_packed(c, s->stream, s->index, s->ws, s->d_rows.p, (int)W, s->d_lens.p, nf, s->ws.spans.p, !spans_mode, s->h_entries.p))) return rc;
    }
    if (ng) {
        if ((rc = run_generic(c, *s, ng, gbytes, gmax, !spans_mode, spans_mode))) return rc;
    }
    if (spans_mode) {
        if (nf) {
            if ((rc = s->h_spans.ensure(nf))) return rc;
            HIP_TRY(hipMemcpyAsync(s->h_spans.p, s->ws.spans.p, nf * sizeof(short2), hipMemcpyDeviceToHost, s->stream));
        }
        if (ng) {
            if ((rc = s->h_gstart.ensure(ng))) return rc;
            if ((rc = s->h_gend.ensure(ng))) return rc;
            HIP_TRY(hipMemcpyAsync(s->h_gstart.p, s->d_gstart.p, ng * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipMemcpyAsync(s->h_gend.p, s->d_gend.p, ng * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
        }
        HIP_TRY(hipStreamSynchronize(s->stream));
        for (uint64_t k = 0; k < nf; ++k) {
            const short2 v = s->h_spans.p[k];
            starts_out[s->fast_idx[k]] = v.x >= 0 ? v.x : -1;
            ends_out[s->fast_idx[k]] = v.y >= 0 ? v.y : -1;
        }
        for (uint64_t k = 0; k < ng; ++k) {
            starts_out[s->gen_idx[k]] = s->h_gstart.p[k];
            ends_out[s->gen_idx[k]] = s->h_gend.p[k];
        }
    } else {
        HIP_TRY(hipEventRecord(s->done, s->stream));
        capacity_note_enqueued(c, s, n);
    }
    return VF_OK;
}

static int submit_reads(vf_ctx *c, const uint8_t *base, const ReadRef *reads, uint64_t n, int32_t *starts, int32_t *ends) {
    for (uint64_t lo = 0; lo < n; lo += c->batch_reads) {
        const uint64_t m = std::min<uint64_t>(c->batch_reads, n - lo);
        int rc = submit_chunk(c, base, reads + lo, m, starts ? starts + lo : nullptr, ends ? ends + lo : nullptr);
        if (rc) return rc;
    }
    return VF_OK;
}

// ------------------------------------------------------------------------------------------------
// public submission entry points
// ------------------------------------------------------------------------------------------------
static int to_refs(const uint64_t *offsets, uint64_t n, std::vector<ReadRef> &out) {
    out.resize(n);
    for (uint64_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return fail(VF_EINVAL, "offsets must be non-decreasing");
        const uint64_t len = offsets[i + 1] - offsets[i];
        if (len > 0xFFFFFFFFull) return fail(VF_EINVAL, "read longer than 4 GiB");
        out[i] = ReadRef{offsets[i], (uint32_t)len};
    }
    return VF_OK;
}

static int vf_submit_ascii_impl(vf_ctx *c, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_reads);
extern "C" int vf_submit_ascii(vf_ctx *c, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_reads) { return vf_guard([&] { return vf_submit_ascii_impl(c, bytes, offsets, n_reads); }); }
static int vf_submit_ascii_impl(vf_ctx *c, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_reads) {
    if (!c || (n_reads && (!offsets || (!bytes && offsets[n_reads] != 0)))) return fail(VF_EINVAL, "null argument");
    std::vector<ReadRef> refs;
    int rc = to_refs(offsets, n_reads, refs);
    if (rc) return rc;
    return submit_reads(c, bytes, refs.data(), n_reads, nullptr, nullptr);
}

static int vf_spans_ascii_impl(vf_ctx *c, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_reads, int32_t *starts, int32_t *ends);
extern "C" int vf_spans_ascii(vf_ctx *c, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_reads, int32_t *starts, int32_t *ends) { return vf_guard([&] { return vf_spans_ascii_impl(c, bytes, offsets, n_reads, starts, ends); }); }
static int vf_spans_ascii_impl(vf_ctx *c, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_reads, int32_t *starts, int32_t *ends) {
    if (!c || !starts || !ends || (n_reads && !offsets)) return fail(VF_EINVAL, "null argument");
    std::vector<ReadRef> refs;
    int rc = to_refs(offsets, n_reads, refs);
    if (rc) return rc;
    return submit_reads(c, bytes, refs.data(), n_reads, starts, ends);
}

static int vf_submit_packed_impl(vf_ctx *c, const uint32_t *rows, uint32_t W, const uint16_t *lens, uint64_t n);
extern "C" int vf_submit_packed(vf_ctx *c, const uint32_t *rows, uint32_t W, const uint16_t *lens, uint64_t n) { return vf_guard([&] { return vf_submit_packed_impl(c, rows, W, lens, n); }); }
static int vf_submit_packed_impl(vf_ctx *c, const uint32_t *rows, uint32_t W, const uint16_t *lens, uint64_t n) {
    if (!c || (n && (!rows || !lens))) return fail(VF_EINVAL, "null argument");
    if (!c->fast_ok) return fail(VF_EINVAL, "these adapters/scores need the byte-wise path: use vf_submit_ascii");
    if (W != 10 && W != 20 && W != 40) return fail(VF_EINVAL, "row_dwords must be 10, 20 or 40");
    HIP_TRY(hipSetDevice(c->dev));
    for (uint64_t lo = 0; lo < n; lo += c->batch_reads) {
        const uint64_t m = std::min<uint64_t>(c->batch_reads, n - lo);
        uint16_t mx = 0;
        for (uint64_t i = 0; i < m; ++i) mx = lens[lo + i] > mx ? lens[lo + i] : mx;      // (vectorised: see vf_batch_submit)
        if (mx > W * 16u) return fail(VF_EINVAL, "read length exceeds row");
        Slot *s = acquire_slot(c);
        struct Rel { vf_ctx *c; Slot *s; ~Rel() { release_slot(c, s); } } rel{c, s};
        int rc;
        if ((rc = wait_slot(c, s))) return rc;
        CapacityTicket ticket{c};
        uint64_t need = 0;
        if (arena_need(c, mx) != 0) for (uint64_t i = 0; i < m; ++i) need += arena_need(c, lens[lo + i]);
        if ((rc = reserve_capacity(c, ticket, m, need))) return rc;
        std::shared_lock<std::shared_mutex> table_lock(c->table_mu);
        if ((rc = s->h_rows.ensure((size_t)m * W))) return rc;
        if ((rc = s->h_lens.ensure(m))) return rc;
        // staging copy split over the host workers (one thread moves ~10 GB/s, PCIe Gen5 x16 takes 63)
        {
            const size_t bytes = (size_t)m * W * 4;
            const uint32_t nt = (uint32_t)std::max<size_t>(1, std::min<size_t>(c->n_threads, bytes >> 22));
            std::vector<std::thread> th;
            for (uint32_t k = 1; k < nt; ++k)
                th.emplace_back([&, k] { memcpy((uint8_t *)s->h_rows.p + bytes * k / nt, (const uint8_t *)(rows + lo * W) + bytes * k / nt, bytes * (k + 1) / nt - bytes * k / nt); });
            memcpy(s->h_rows.p, rows + lo * W, bytes / nt);
            memcpy(s->h_lens.p, lens + lo, m * sizeof(uint16_t));
            for (auto &t : th) t.join();
        }
        if ((rc = s->d_rows.ensure((size_t)m * W))) return rc;
        if ((rc = s->d_lens.ensure(m))) return rc;
        if ((rc = s->ws.ensure(m, (int)W))) return rc;
        HIP_TRY(hipMemcpyAsync(s->d_rows.p, s->h_rows.p, (size_t)m * W * 4, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(s->d_lens.p, s->h_lens.p, m * sizeof(uint16_t), hipMemcpyHostToDevice, s->stream));
        if ((rc = run_packed(c, s->stream, s->index, s->ws, s->d_rows.p, (int)W, s->d_lens.p, m, s->ws.spans.p, true, s->h_entries.p))) return rc;
        HIP_TRY(hipEventRecord(s->done, s->stream));
        capacity_note_enqueued(c, s, m);
        c->n_reads += m;
    }
    return VF_OK;
}

// ------------------------------------------------------------------------------------------------
// pinned batches: the caller decodes straight into a slot's page-locked staging (north_star: "decoded on the host into
// pinned packed-base buffers and streamed to HBM via hipMemcpyAsync overlapped with the previous batch").
// vf_submit_packed takes ordinary host memory and has to copy it into that staging first: 22 GB/s instead of the link's rate.
// ------------------------------------------------------------------------------------------------
static int vf_batch_acquire_impl(vf_ctx *c, uint64_t n_reads, uint32_t W, vf_batch **out, uint32_t **rows, uint16_t **lens);
extern "C" int vf_batch_acquire(vf_ctx *c, uint64_t n_reads, uint32_t W, vf_batch **out, uint32_t **rows, uint16_t **lens) { return vf_guard([&] { return vf_batch_acquire_impl(c, n_reads, W, out, rows, lens); }); }
static int vf_batch_acquire_impl(vf_ctx *c, uint64_t n_reads, uint32_t W, vf_batch **out, uint32_t **rows, uint16_t **lens) {
    if (!c || !out || !rows || !lens) return fail(VF_EINVAL, "null argument");
    *out = nullptr;
    if (!c->fast_ok) return fail(VF_EINVAL, "these adapters/scores need the byte-wise path: use vf_submit_ascii");
    if (W != 10 && W != 20 && W != 40) return fail(VF_EINVAL, "row_dwords must be 10, 20 or 40");
    if (n_reads == 0 || n_reads > 0xFFFFFFF0ull) return fail(VF_EINVAL, "n_reads out of range");
    HIP_TRY(hipSetDevice(c->dev));
    Slot *s = acquire_slot(c);                                  // owned by the caller until vf_batch_submit / _release
    int rc;
    if ((rc = wait_slot(c, s)) || (rc = s->h_rows.ensure((size_t)n_reads * W)) || (rc = s->h_lens.ensure(n_reads))) { release_slot(c, s); return rc; }
    { std::lock_guard<std::mutex> g(c->mu); s->batch_cap = n_reads; s->batch_W = W; }
    *rows = s->h_rows.p; *lens = s->h_lens.p;
    *out = reinterpret_cast<vf_batch *>(s);
    return VF_OK;
}
static int vf_batch_release_impl(vf_ctx *c, vf_batch *b);
extern "C" int vf_batch_release(vf_ctx *c, vf_batch *b) { return vf_guard([&] { return vf_batch_release_impl(c, b); }); }
// A batch handle is honoured once: it must be a slot of THIS context that vf_batch_acquire handed out and nobody has given
// back yet.  A stale or doubly submitted handle used to release a slot another thread might just have acquired - two
// writers on one pinned staging buffer and its stream.  Returns the slot, now owned by the caller, and the reads it was
// acquired for; nullptr for any other handle (nothing touched).
static Slot *batch_take(vf_ctx *c, vf_batch *b, uint64_t *cap) {
    std::lock_guard<std::mutex> g(c->mu);
    for (auto &x : c->slots)
        if (reinterpret_cast<vf_batch *>(x.get()) == b) {
            if (!x->busy || x->batch_cap == 0) return nullptr;
            *cap = x->batch_cap;
            x->batch_cap = 0;
            return x.get();
        }
    return nullptr;
}
static int vf_batch_release_impl(vf_ctx *c, vf_batch *b) {
    if (!c || !b) return fail(VF_EINVAL, "null argument");
    uint64_t cap = 0;
    Slot *s = batch_take(c, b, &cap);
    if (!s) return fail(VF_EINVAL, "not a batch handle of this context that is currently handed out");
    release_slot(c, s);
    return VF_OK;
}
static int vf_batch_submit_impl(vf_ctx *c, vf_batch *b, uint64_t n);
extern "C" int vf_batch_submit(vf_ctx *c, vf_batch *b, uint64_t n) { return vf_guard([&] { return vf_batch_submit_impl(c, b, n); }); }
static int vf_batch_submit_impl(vf_ctx *c, vf_batch *b, uint64_t n) {
    if (!c || !b) return fail(VF_EINVAL, "null argument");
    uint64_t cap = 0;
    Slot *s = batch_take(c, b, &cap);
    if (!s) return fail(VF_EINVAL, "not a batch handle of this context that is currently handed out");
    struct Rel { vf_ctx *c; Slot *s; ~Rel() { release_slot(c, s); } } rel{c, s};      // (armed only for a handle that was ours to take)
    if (n > cap) return fail(VF_EINVAL, "more reads than the batch was acquired for");
    if (n == 0) return VF_OK;
    const uint32_t W = s->batch_W;
    HIP_TRY(hipSetDevice(c->dev));
    // longest read first, in a loop the compiler vectorises: the per-read form of this check (a call and two branches a
    // read) took 0.5 ms of host time per 1.25 M-read batch - more than half the batch's copy over the link, and the single
    // submitting thread of the PCIe-inclusive leg paced the link at 30 of 56 GB/s (profiles/r03_pcie_timeline.md)
    uint64_t need = 0;
    {
        const uint16_t *hl = s->h_lens.p;
        uint16_t mx = 0;
        for (uint64_t i = 0; i < n; ++i) mx = hl[i] > mx ? hl[i] : mx;
        if (mx > W * 16u) return fail(VF_EINVAL, "read length exceeds row");
        if (arena_need(c, mx) != 0)                       // only reads long enough for an arena key cost anything
            for (uint64_t i = 0; i < n; ++i) need += arena_need(c, hl[i]);
    }
    int rc;
    static const bool dbg_t = [] { const char *v = getenv("VFIND_DEBUG_SUBMIT"); return v && atoi(v) != 0; }();      // host time of each phase, stderr
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = dbg_t ? now() : 0;
    CapacityTicket ticket{c};
    if ((rc = reserve_capacity(c, ticket, n, need))) return rc;
    const double t1 = dbg_t ? now() : 0;
    std::shared_lock<std::shared_mutex> table_lock(c->table_mu);
    if ((rc = s->d_rows.ensure((size_t)n * W)) || (rc = s->d_lens.ensure(n)) || (rc = s->ws.ensure(n, (int)W))) return rc;
    const double t2 = dbg_t ? now() : 0;
    HIP_TRY(hipMemcpyAsync(s->d_rows.p, s->h_rows.p, (size_t)n * W * 4, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->d_lens.p, s->h_lens.p, n * sizeof(uint16_t), hipMemcpyHostToDevice, s->stream));
    const double t3 = dbg_t ? now() : 0;
    if ((rc = run_packed(c, s->stream, s->index, s->ws, s->d_rows.p, (int)W, s->d_lens.p, n, s->ws.spans.p, true, s->h_entries.p))) return rc;
    if (dbg_t) fprintf(stderr, "[submit] capacity %.2f ensure %.2f copies %.2f kernels %.2f ms\n", t1 - t0, t2 - t1, t3 - t2, now() - t3);
    HIP_TRY(hipEventRecord(s->done, s->stream));
    capacity_note_enqueued(c, s, n);
    c->n_reads += n;
    return VF_OK;
}

static int vf_process_device_packed_impl(vf_ctx *c, const void *d_rows, uint32_t W, const void *d_lens, uint64_t n, void *d_spans);
extern "C" int vf_process_device_packed(vf_ctx *c, const void *d_rows, uint32_t W, const void *d_lens, uint64_t n, void *d_spans) { return vf_guard([&] { return vf_process_device_packed_impl(c, d_rows, W, d_lens, n, d_spans); }); }
// reads per launch set of the resident path (VFIND_RES_SPLIT; 0 = a call's reads in one piece)
static uint64_t res_split_reads() {
    static const uint64_t v = [] { const char *e = getenv("VFIND_RES_SPLIT"); const long long x = e ? atoll(e) : 0; return (uint64_t)(x > 0 ? std::max<long long>(x, 65536) : 0); }();
    return v;
}
static int process_device_piece(vf_ctx *c, const uint32_t *d_rows, uint32_t W, const uint16_t *d_lens, uint64_t n, short2 *d_spans);
static int vf_process_device_packed_impl(vf_ctx *c, const void *d_rows, uint32_t W, const void *d_lens, uint64_t n, void *d_spans) {
    if (!c || (n && (!d_rows || !d_lens))) return fail(VF_EINVAL, "null argument");
    if (!c->fast_ok) return fail(VF_EINVAL, "these adapters/scores need the byte-wise path: use vf_submit_ascii");
    if (W != 10 && W != 20 && W != 40) return fail(VF_EINVAL, "row_dwords must be 10, 20 or 40");
    HIP_TRY(hipSetDevice(c->dev));
    // A resident batch may be cut into pieces that go down the batch lanes one after the other: a piece's rows are read by
    // k_scan, gathered from by the filter and DP kernels and streamed again by k_finish - pieces small enough for the rows of
    // the pieces in flight (and the table) to stay in the 256 MiB Infinity Cache make every visit after the first an on-die hit.
    uint64_t piece = res_split_reads();
    if (piece == 0 && n > (1ull << 30)) piece = 1ull << 30;          // (a launch set takes at most 2^31 reads: 31-bit task entries)
    if (piece == 0 || n <= piece + piece / 2) return process_device_piece(c, (const uint32_t *)d_rows, W, (const uint16_t *)d_lens, n, (short2 *)d_spans);
    const uint64_t k = (n + piece - 1) / piece, each = ((n + k - 1) / k + 255) & ~255ull;      // equal pieces, whole workgroups
    for (uint64_t at = 0; at < n; at += each) {
        const uint64_t m = std::min(each, n - at);
        const int rc = process_device_piece(c, (const uint32_t *)d_rows + at * W, W, (const uint16_t *)d_lens + at, m, d_spans ? (short2 *)d_spans + at : nullptr);
        if (rc) return rc;
    }
    return VF_OK;
}
static int process_device_piece(vf_ctx *c, const uint32_t *d_rows, uint32_t W, const uint16_t *d_lens, uint64_t n, short2 *d_spans) {
    int rc;
    ResLane &L = c->res[c->res_next++ % c->n_res];
    if (L.in_flight && hipEventQuery(L.done) == hipSuccess) {           // lane idle: its entry count is current
        if ((rc = table_grow_if_needed(c, *L.h_entries.p))) return rc;
    }
    std::shared_lock<std::shared_mutex> table_lock(c->table_mu);
    if ((rc = L.ws.ensure(n, (int)W))) return rc;
    short2 *spans = d_spans ? d_spans : L.ws.spans.p;
    if ((rc = run_packed(c, L.stream, L.chunk_row, L.ws, d_rows, (int)W, d_lens, n, spans, true, L.h_entries.p))) return rc;
    HIP_TRY(hipEventRecord(L.done, L.stream));
    L.inflight_reads = n;
    L.in_flight = true;
    c->n_reads += n;
    // diagnosis: tasks per filter class of this batch (synchronises the lane; never set in a measurement)
    static const bool dbg_classes = [] { const char *v = getenv("VFIND_DEBUG_CLASSES"); return v && atoi(v) != 0; }();
    if (dbg_classes) {
        uint32_t tot[16];
        HIP_TRY(hipStreamSynchronize(L.stream));
        HIP_TRY(hipMemcpy(tot, L.ws.cnt.p + 10 * kMaxGrid, sizeof(tot), hipMemcpyDeviceToHost));
        fprintf(stderr, "vfind classes (narrow band, banded, wide band, window, long window, two windows, full width): prefix %u %u %u %u %u %u %u  suffix %u %u %u %u %u %u %u\n",
                tot[5], tot[0], tot[6], tot[1], tot[2], tot[3], tot[4], tot[13], tot[8], tot[14], tot[9], tot[10], tot[11], tot[12]);
        {
            uint64_t hits = 0;
            HIP_TRY(hipMemcpy(&hits, c->t_counters.p + C_DBG4, sizeof(hits), hipMemcpyDeviceToHost));
            if (hits) fprintf(stderr, "vfind table stage: %llu reads counted in a key cache so far (diagnosis build)\n", (unsigned long long)hits);
        }
    }
    return VF_OK;
}

static int check_device_errors(vf_ctx *c, uint64_t *counters_out) {
    uint64_t h[C_NUM];
    HIP_TRY(hipMemcpy(h, c->t_counters.p, sizeof(h), hipMemcpyDeviceToHost));
    if (counters_out) memcpy(counters_out, h, sizeof(h));
    if (h[C_ERROR] & 1) return fail(VF_ECAPACITY, "device key arena full: raise arena_bytes");
    if (h[C_ERROR] & 2) return fail(VF_ECAPACITY, "device count table full: raise table_slots");
    if (h[C_ERROR] & 4) return fail(VF_ECAPACITY, "a variant region exceeds 65535 symbols");
    if (h[C_ERROR] & 8) return fail(VF_EDEVICE, "count log: part totals and entries disagree (internal error)");
    if (h[C_ERROR] & 16) return fail(VF_EDEVICE, "key cache: occurrences of a key whose insert failed (internal error)");
    if (h[C_ERROR] & 32) return fail(VF_EINVAL, "merge: the blob holds byte keys with non-ASCII bytes, which a translating context's table cannot store (was it exported with skip_translation?)");
    return VF_OK;
}

static int vf_sync_impl(vf_ctx *c);
extern "C" int vf_sync(vf_ctx *c) { return vf_guard([&] { return vf_sync_impl(c); }); }
static int vf_sync_impl(vf_ctx *c) {
    if (!c) return fail(VF_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->dev));
    {
        std::lock_guard<std::mutex> g(c->err_mu);
        if (c->async_rc) { int rc = c->async_rc; std::string m = c->async_msg; c->async_rc = 0; return fail(rc, m); }
    }
    for (auto &s : c->slots) {
        // a slot is only touched by its owner between acquire/release; here nobody may be submitting
        HIP_TRY(hipStreamSynchronize(s->stream));
        capacity_retire(c, s.get());
        s->in_flight = false;
        (void)collect_text_result(c, s.get());
    }
    for (auto &L : c->res) {
        if (!L.stream) continue;
        HIP_TRY(hipStreamSynchronize(L.stream));
        L.in_flight = false;
    }
    HIP_TRY(hipStreamSynchronize(c->main_stream));
    resolve_events(c);
    {
        std::lock_guard<std::mutex> g(c->err_mu);
        if (c->async_rc) { int rc = c->async_rc; std::string m = c->async_msg; c->async_rc = 0; return fail(rc, m); }
    }
    return check_device_errors(c, nullptr);
}

// ------------------------------------------------------------------------------------------------
// table: clear / export / merge / finish
// ------------------------------------------------------------------------------------------------
// Asynchronous: ordered after every batch already submitted (device-side event waits) and before
// everything submitted afterwards; the host does not block.
static int vf_table_clear_impl(vf_ctx *c);
extern "C" int vf_table_clear(vf_ctx *c) { return vf_guard([&] { return vf_table_clear_impl(c); }); }
static int vf_table_clear_impl(vf_ctx *c) {
    if (!c) return fail(VF_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->dev));
    for (auto &s : c->slots)
        if (s->in_flight) HIP_TRY(hipStreamWaitEvent(c->main_stream, s->done, 0));
    for (auto &L : c->res)
        if (L.in_flight) HIP_TRY(hipStreamWaitEvent(c->main_stream, L.done, 0));
    std::unique_lock<std::shared_mutex> table_lock(c->table_mu);        // the table in use changes hands below
    {
        std::lock_guard<std::mutex> g(c->cap_mu);
        c->upper_entries = c->unenqueued; c->upper_arena = c->unenqueued_arena;
        c->known_entries = 0; c->pending_reads = 0;                       // (batches still in flight count into the table being dropped)
        for (auto &s : c->slots) s->cap_pending = false;
    }
    for (auto &s : c->slots) if (s->h_entries.p) *s->h_entries.p = 0;  // (growth hints of the table being dropped)
    for (auto &L : c->res) if (L.h_entries.p) *L.h_entries.p = 0;
    c->part_world = 0;
    // The switch pays while the clear is small beside a batch: 2^21 slots (268 MB to zero) -3 % per step of cfg 2; 2^25 slots
    // (4.3 GB, the cfg-5 shape at 10 M reads a step) +8 % - that clear is better off alone on the stream than beside
    // the kernels of the next batch for a whole millisecond.
    constexpr uint64_t kSpareMaxSlots = 1ull << 22;
    const bool small = c->table_slots <= kSpareMaxSlots && (!c->spare.valid || c->spare.table_slots <= kSpareMaxSlots);
    if (c->spare_on && small && !c->spare.valid) {
        int rc = table_make_spare(c);
        if (rc == VF_ENOMEM) c->spare_on = false;          // no room for a second table: clear in place, as before
        else if (rc) return rc;
    }
    int rc = table_clear_locked(c);
    if (rc == VF_OK && c->spare.valid && small) table_swap_spare(c);
    return rc;
}

static int vf_wait_stream_impl(vf_ctx *c, void *hip_stream);
extern "C" int vf_wait_stream(vf_ctx *c, void *hip_stream) { return vf_guard([&] { return vf_wait_stream_impl(c, hip_stream); }); }
static int vf_wait_stream_impl(vf_ctx *c, void *hip_stream) {
    if (!c) return fail(VF_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->dev));
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    struct Drop { hipEvent_t e; ~Drop() { (void)hipEventDestroy(e); } } drop{ev};      // (destroyed once the waits below have consumed it)
    HIP_TRY(hipEventRecord(ev, (hipStream_t)hip_stream));
    HIP_TRY(hipStreamWaitEvent(c->main_stream, ev, 0));
    for (auto &L : c->res) if (L.stream) HIP_TRY(hipStreamWaitEvent(L.stream, ev, 0));      // vf_process_device_packed reads caller memory on these
    return VF_OK;
}

static uint64_t blob_bytes(uint64_t n, uint64_t arena_dw) { return 16 + n * 16 + arena_dw * 4; }

static int vf_table_export_size_impl(vf_ctx *c, uint64_t *nbytes);
extern "C" int vf_table_export_size(vf_ctx *c, uint64_t *nbytes) { return vf_guard([&] { return vf_table_export_size_impl(c, nbytes); }); }
static int vf_table_export_size_impl(vf_ctx *c, uint64_t *nbytes) {
    if (!c || !nbytes) return fail(VF_EINVAL, "null argument");
    int rc = vf_sync(c);
    if (rc) return rc;
    uint64_t h[C_NUM];
    if ((rc = check_device_errors(c, h))) return rc;
    *nbytes = blob_bytes(h[C_ENTRIES], h[C_KEY_DWORDS]);
    return VF_OK;
}

// blob = {u64 n, u64 arena_dwords, u64 refs[n], u64 counts[n], u32 arena[arena_dwords]}
// G = 0: the whole table (n, adw = its entry and key-dword counters); otherwise owner's share of G (n, adw from k_owner_count)
static int export_blob(vf_ctx *c, uint32_t G, uint32_t owner, uint64_t n, uint64_t adw, void *d_dst, uint64_t capacity, uint64_t *written) {
    const uint64_t need = blob_bytes(n, adw);
    if (capacity < need) return fail(VF_EINVAL, "export buffer too small");
    uint8_t *dst = (uint8_t *)d_dst;
    const uint64_t hdr[2] = {n, adw};
    HIP_TRY(hipMemcpyAsync(dst, hdr, 16, hipMemcpyHostToDevice, c->main_stream));
    HIP_TRY(hipMemsetAsync(c->t_counters.p + C_EXPORT_N, 0, 2 * sizeof(uint64_t), c->main_stream));
    HIP_TRY(launch_export(c->main_stream, c->tab, (uint64_t *)(dst + 16), (uint64_t *)(dst + 16 + n * 8), (uint32_t *)(dst + 16 + n * 16), G, owner));
    HIP_TRY(hipStreamSynchronize(c->main_stream));
    uint64_t got[2] = {0, 0};
    HIP_TRY(hipMemcpy(got, c->t_counters.p + C_EXPORT_N, 16, hipMemcpyDeviceToHost));
    if (got[0] != n || got[1] != adw) return fail(VF_EDEVICE, "table export count mismatch");
    if (written) *written = need;
    return VF_OK;
}

static int vf_table_export_impl(vf_ctx *c, void *d_dst, uint64_t capacity, uint64_t *written);
extern "C" int vf_table_export(vf_ctx *c, void *d_dst, uint64_t capacity, uint64_t *written) { return vf_guard([&] { return vf_table_export_impl(c, d_dst, capacity, written); }); }
static int vf_table_export_impl(vf_ctx *c, void *d_dst, uint64_t capacity, uint64_t *written) {
    if (!c || !d_dst) return fail(VF_EINVAL, "null argument");
    int rc = vf_sync(c);
    if (rc) return rc;
    uint64_t h[C_NUM];
    if ((rc = check_device_errors(c, h))) return rc;
    return export_blob(c, 0, 0, h[C_ENTRIES], h[C_KEY_DWORDS], d_dst, capacity, written);
}

// ---- owner-partitioned exchange: every key has one owner among the `world` GPUs ---------------------------
// entries and key dwords of every owner's share of this GPU's table (cached in the context for the exports that follow)
static int partition_counts(vf_ctx *c, uint32_t world) {
    if (world == 0 || world > 64) return fail(VF_EINVAL, "world must be 1..64");
    int rc = vf_sync(c);
    if (rc) return rc;
    if ((rc = check_device_errors(c, nullptr))) return rc;
    if ((rc = c->d_part.ensure(2 * 64))) return rc;
    HIP_TRY(hipMemsetAsync(c->d_part.p, 0, 2 * 64 * sizeof(uint64_t), c->main_stream));
    HIP_TRY(launch_owner_count(c->main_stream, c->tab, world, c->d_part.p));
    HIP_TRY(hipMemcpyAsync(c->part, c->d_part.p, 2 * (size_t)world * sizeof(uint64_t), hipMemcpyDeviceToHost, c->main_stream));
    HIP_TRY(hipStreamSynchronize(c->main_stream));
    c->part_world = world;
    return VF_OK;
}
static int vf_table_partition_sizes_impl(vf_ctx *c, uint32_t world, uint64_t *nbytes);
extern "C" int vf_table_partition_sizes(vf_ctx *c, uint32_t world, uint64_t *nbytes) { return vf_guard([&] { return vf_table_partition_sizes_impl(c, world, nbytes); }); }
static int vf_table_partition_sizes_impl(vf_ctx *c, uint32_t world, uint64_t *nbytes) {
    if (!c || !nbytes) return fail(VF_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->dev));
    int rc = partition_counts(c, world);
    if (rc) return rc;
    for (uint32_t o = 0; o < world; ++o) nbytes[o] = blob_bytes(c->part[2 * o], c->part[2 * o + 1]);
    return VF_OK;
}
static int vf_table_partition_export_impl(vf_ctx *c, uint32_t world, uint32_t owner, void *d_dst, uint64_t capacity, uint64_t *written);
extern "C" int vf_table_partition_export(vf_ctx *c, uint32_t world, uint32_t owner, void *d_dst, uint64_t capacity, uint64_t *written) { return vf_guard([&] { return vf_table_partition_export_impl(c, world, owner, d_dst, capacity, written); }); }
static int vf_table_partition_export_impl(vf_ctx *c, uint32_t world, uint32_t owner, void *d_dst, uint64_t capacity, uint64_t *written) {
    if (!c || !d_dst) return fail(VF_EINVAL, "null argument");
    if (world == 0 || owner >= world) return fail(VF_EINVAL, "owner out of range");
    HIP_TRY(hipSetDevice(c->dev));
    if (c->part_world != world) { int rc = partition_counts(c, world); if (rc) return rc; }
    return export_blob(c, world, owner, c->part[2 * owner], c->part[2 * owner + 1], d_dst, capacity, written);
}

// The 16-byte header of a blob the CALLER owns, read in stream order on the main stream - behind whatever vf_wait_stream made
// a dependency of it (a collective still writing the buffer on a stream of the caller's) - through a one-wave kernel into
// page-locked memory.  A blocking hipMemcpy on the null stream is ordered with none of that on a non-blocking stream: it
// could bring back a stale header, and a wrong or zero entry count loses counts silently (the bug class vf_wait_stream was
// added to close).
static int read_blob_header(vf_ctx *c, const void *d_blob, uint64_t hdr[2]) {
    int rc;
    if ((rc = c->h_hdr.ensure(2))) return rc;
    HIP_TRY(launch_peek(c->main_stream, d_blob, c->h_hdr.p, 4));
    HIP_TRY(hipStreamSynchronize(c->main_stream));
    hdr[0] = c->h_hdr.p[0]; hdr[1] = c->h_hdr.p[1];
    return VF_OK;
}

// header of an exported table as a peer sent it: checked without overflow before anything is sized from it
static int blob_header_ok(const uint64_t hdr[2], uint64_t nbytes) {
    if (nbytes < 16) return fail(VF_EINVAL, "blob too small");
    const uint64_t room = nbytes - 16;
    if (hdr[0] > room / 16 || hdr[1] > (room - hdr[0] * 16) / 4) return fail(VF_EINVAL, "blob truncated");
    return VF_OK;
}

static int vf_table_merge_impl(vf_ctx *c, const void *d_blob, uint64_t nbytes);
extern "C" int vf_table_merge(vf_ctx *c, const void *d_blob, uint64_t nbytes) { return vf_guard([&] { return vf_table_merge_impl(c, d_blob, nbytes); }); }
static int vf_table_merge_impl(vf_ctx *c, const void *d_blob, uint64_t nbytes) {
    if (!c || !d_blob) return fail(VF_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->dev));
    if (nbytes < 16) return fail(VF_EINVAL, "blob too small");
    uint64_t hdr[2];
    int rc;
    if ((rc = read_blob_header(c, d_blob, hdr))) return rc;
    if ((rc = blob_header_ok(hdr, nbytes))) return rc;
    // every peer entry may be a new key (and every key dword may belong to a LONG key): make room first, exactly as a
    // streaming submit does -- a table sized for one rank's keys would otherwise fill up under seven peers' tables
    {
        uint64_t cur;
        { std::lock_guard<std::mutex> g(c->cap_mu); cur = c->upper_entries; }
        if ((cur + hdr[0]) * 2 > c->table_slots || hdr[1]) {
            if ((rc = table_grow_locked(c, false, hdr[0], hdr[1]))) return rc;
        }
        std::lock_guard<std::mutex> g(c->cap_mu);
        c->upper_entries += hdr[0]; c->upper_arena += hdr[1];
        c->known_entries += hdr[0];
    }
    std::shared_lock<std::shared_mutex> table_lock(c->table_mu);
    c->part_world = 0;
    const uint8_t *b = (const uint8_t *)d_blob;
    HIP_TRY(launch_merge(c->main_stream, c->tab, (const uint64_t *)(b + 16), (const uint64_t *)(b + 16 + hdr[0] * 8),
                         (const uint32_t *)(b + 16 + hdr[0] * 16), hdr[0]));
    HIP_TRY(hipStreamSynchronize(c->main_stream));
    return check_device_errors(c, nullptr);
}

// ------------------------------------------------------------------------------------------------
// RCCL (the multi-GPU reduce of the count table over xGMI).  librccl is loaded on first use: a host that runs one GPU
// never needs it, and inside a process that already holds an RCCL (PyTorch ships one under the same soname) the loader
// hands back that copy.
// ------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                          // (optional: absent in very old builds)
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;  // (optional)
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so"}) { r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
        if (!r.h) return;
        auto sym = [&](const char *n) { return dlsym(r.h, n); };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
        r.CommGetAsyncError = (decltype(r.CommGetAsyncError))sym("ncclCommGetAsyncError");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) r.h = nullptr;
    });
    return r.h ? &r : nullptr;
}
}  // namespace
#define NCCL_TRY(expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess)                                                                             \
            return fail(VF_EDEVICE, std::string(#expr) + ": " + (R->GetErrorString ? R->GetErrorString(_r) : "RCCL error")); \
    } while (0)

// test hook: VFIND_TEST_FAIL_ALLREDUCE=<rank>:<step> makes that rank fail at local step 1 (buckets), 2 (buffers) or 3 (share),
// or - step 4 - in a send INSIDE the first round's begin/end bracket (LocalTransport)
static bool allreduce_test_fail(int rank, int step) {
    const char *e = getenv("VFIND_TEST_FAIL_ALLREDUCE");
    int r = -1, s = -1;
    return e && sscanf(e, "%d:%d", &r, &s) == 2 && r == rank && s == step;
}

// ---- transports of the exchange ---------------------------------------------------------------------------------------
// vf_table_allreduce is written against these five calls.  RcclTransport is the product path (one process per GPU, xGMI);
// LocalTransport connects contexts of ONE process - a host that drives several GPUs from one process, and the tests, where
// RCCL refuses two ranks on one device: every byte moves by a device-to-device copy behind a host rendezvous, so the same
// bucket / offset / merge code runs with G > 1 on a one-GPU box.
namespace {
struct Transport {
    virtual ~Transport() {}
    // count 64-bit words from every rank, rank r's at recv + r * count (device pointers; in stream order)
    virtual int all_gather(const uint64_t *send, uint64_t *recv, size_t count, hipStream_t st) = 0;
    // one round of point-to-point transfers: the calls between begin and end are matched with the peers' and complete
    // (in stream order) behind end.  Every rank of the communicator enters every round, with or without transfers.
    virtual int begin() = 0;
    virtual int send(const void *p, size_t bytes, int peer, hipStream_t st) = 0;
    virtual int recv(void *p, size_t bytes, int peer, hipStream_t st) = 0;
    virtual int end(hipStream_t st) = 0;
    // what has been enqueued on `st` has completed, or the transport has given up on it (a peer that failed, a deadline):
    // the wait every rank does behind a collective or a round - never an unbounded hipStreamSynchronize
    virtual int wait(hipStream_t st) = 0;
    // This rank is leaving with an error: a round left open between begin() and end() is closed, and the communicator is
    // broken so that no peer stays blocked on this rank (and this rank on no peer).  Safe to call in any state, once or more;
    // the transport is unusable afterwards.  The caller's error message is kept.
    virtual void abort() = 0;
};

// seconds a rank waits for a collective or a round of the exchange before it gives up on its peers (0 = for ever)
static double exchange_timeout_s() {
    static const double t = [] { const char *e = getenv("VFIND_EXCHANGE_TIMEOUT_S"); const double v = e ? atof(e) : 120.0; return v >= 0. ? v : 120.0; }();
    return t;
}

struct RcclTransport : Transport {
    Rccl *R; ncclComm_t comm; void **owner;      // owner: the context's handle, cleared when the communicator is aborted
    bool in_group = false, dead = false;
    RcclTransport(Rccl *r, ncclComm_t c, void **owner_) : R(r), comm(c), owner(owner_) {}
    int all_gather(const uint64_t *send, uint64_t *recv, size_t count, hipStream_t st) override { NCCL_TRY(R->AllGather(send, recv, count, ncclUint64, comm, st)); return VF_OK; }
    int begin() override { NCCL_TRY(R->GroupStart()); in_group = true; return VF_OK; }
    int send(const void *p, size_t bytes, int peer, hipStream_t st) override { NCCL_TRY(R->Send(p, bytes, ncclUint8, peer, comm, st)); return VF_OK; }
    int recv(void *p, size_t bytes, int peer, hipStream_t st) override { NCCL_TRY(R->Recv(p, bytes, ncclUint8, peer, comm, st)); return VF_OK; }
    int end(hipStream_t) override { in_group = false; NCCL_TRY(R->GroupEnd()); return VF_OK; }
    // The stream is polled, never waited for blindly: a peer that has gone (its process died, it aborted its communicator)
    // leaves this rank's collective kernel spinning for ever.  RCCL reports what it notices through ncclCommGetAsyncError;
    // what it does not notice ends at the deadline.  Either way the communicator is aborted - that is what makes the
    // kernel leave - and the call returns an error.
    int wait(hipStream_t st) override {
        const auto t0 = std::chrono::steady_clock::now();
        const double limit = exchange_timeout_s();
        for (unsigned spin = 0;; ++spin) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return VF_OK;
            if (q != hipErrorNotReady) { (void)hipGetLastError(); const int rc = fail(VF_EDEVICE, std::string("table exchange: ") + hipGetErrorString(q)); abort(); return rc; }
            if (R->CommGetAsyncError && (spin & 63u) == 63u) {
                ncclResult_t ar = ncclSuccess;
                if (R->CommGetAsyncError(comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                    const int rc = fail(VF_EDEVICE, std::string("table exchange: RCCL reports an asynchronous error: ") + (R->GetErrorString ? R->GetErrorString(ar) : "?"));
                    abort();
                    return rc;
                }
            }
            if (limit > 0. && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
                const int rc = fail(VF_EDEVICE, "table exchange: no completion within VFIND_EXCHANGE_TIMEOUT_S (" + std::to_string((int)limit) + " s): a peer has gone; communicator aborted");
                abort();
                return rc;
            }
            if (spin < 2000u) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    void abort() override {
        const std::string keep = g_err;
        if (in_group) { in_group = false; (void)R->GroupEnd(); }       // an open ncclGroupStart would swallow every later call of this thread
        if (!dead) {
            dead = true;
            if (R->CommAbort) (void)R->CommAbort(comm); else (void)R->CommDestroy(comm);
            if (owner) *owner = nullptr;                                // (freed: vf_comm_release must not destroy it again)
        }
        g_err = keep;
    }
};
}  // namespace

// The rendezvous of a LocalTransport group: one per group of contexts, shared by their threads.
struct vf_local_group {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;                                     // a rank left with an error: nobody waits any more
    struct Post { const void *p = nullptr; size_t bytes = 0; int dev = 0; };
    std::vector<Post> gather;                                // [rank]
    std::vector<Post> mail;                                  // [from * world + to]
    // every rank arrives; false = the group is broken (a rank failed or left)
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const uint64_t gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return generation != gen || broken; });
        return !broken;
    }
    void abort() { { std::lock_guard<std::mutex> g(mu); broken = true; } cv.notify_all(); }
};

namespace {
struct LocalTransport : Transport {
    vf_local_group *g; int me, dev;
    struct Op { bool is_send; void *p; size_t bytes; int peer; };
    std::vector<Op> ops;
    LocalTransport(vf_local_group *g_, int rank, int dev_) : g(g_), me(rank), dev(dev_) {}
    int broken() { return fail(VF_EDEVICE, "local exchange: a peer left the exchange with an error"); }
    int copy(void *dst, const vf_local_group::Post &src, size_t bytes, hipStream_t st) {
        if (src.dev == dev) HIP_TRY(hipMemcpyAsync(dst, src.p, bytes, hipMemcpyDeviceToDevice, st));
        else HIP_TRY(hipMemcpyPeerAsync(dst, dev, src.p, src.dev, bytes, st));
        return VF_OK;
    }
    int all_gather(const uint64_t *send, uint64_t *recv, size_t count, hipStream_t st) override {
        HIP_TRY(hipStreamSynchronize(st));                   // what `send` holds was written in stream order
        { std::lock_guard<std::mutex> lk(g->mu); g->gather[me] = {send, count * sizeof(uint64_t), dev}; }
        if (!g->barrier()) return broken();
        int rc = VF_OK;
        for (int r = 0; r < g->world && !rc; ++r) {
            vf_local_group::Post src;
            { std::lock_guard<std::mutex> lk(g->mu); src = g->gather[r]; }
            if (src.bytes != count * sizeof(uint64_t)) rc = fail(VF_EDEVICE, "local exchange: all_gather counts differ between ranks");
            else rc = copy(recv + (size_t)r * count, src, src.bytes, st);
        }
        if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = fail(VF_EDEVICE, "hipStreamSynchronize failed");
        if (rc) { g->abort(); return rc; }
        if (!g->barrier()) return broken();                  // nobody reuses its send buffer before everyone has copied
        return VF_OK;
    }
    int wait(hipStream_t st) override { HIP_TRY(hipStreamSynchronize(st)); return VF_OK; }     // (device-to-device copies of this process)
    void abort() override { const std::string keep = g_err; ops.clear(); g->abort(); g_err = keep; }   // peers waiting at the rendezvous leave too
    int begin() override { ops.clear(); return VF_OK; }
    int send(const void *p, size_t bytes, int peer, hipStream_t) override {
        if (allreduce_test_fail(me, 4)) return fail(VF_EDEVICE, "injected failure inside a round (VFIND_TEST_FAIL_ALLREDUCE)");
        ops.push_back({true, const_cast<void *>(p), bytes, peer}); return VF_OK;
    }
    int recv(void *p, size_t bytes, int peer, hipStream_t) override { ops.push_back({false, p, bytes, peer}); return VF_OK; }
    int end(hipStream_t st) override {
        HIP_TRY(hipStreamSynchronize(st));
        {
            std::lock_guard<std::mutex> lk(g->mu);
            for (const Op &o : ops) if (o.is_send) g->mail[(size_t)me * g->world + o.peer] = {o.p, o.bytes, dev};
        }
        if (!g->barrier()) return broken();
        int rc = VF_OK;
        for (const Op &o : ops) {
            if (o.is_send || rc) continue;
            vf_local_group::Post src;
            { std::lock_guard<std::mutex> lk(g->mu); src = g->mail[(size_t)o.peer * g->world + me]; }
            if (!src.p || src.bytes != o.bytes) rc = fail(VF_EDEVICE, "local exchange: a receive has no matching send of the same size");
            else rc = copy(o.p, src, o.bytes, st);
        }
        if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = fail(VF_EDEVICE, "hipStreamSynchronize failed");
        if (rc) { g->abort(); return rc; }
        if (!g->barrier()) return broken();
        {
            std::lock_guard<std::mutex> lk(g->mu);
            for (const Op &o : ops) if (o.is_send) g->mail[(size_t)me * g->world + o.peer] = {};
        }
        ops.clear();
        return VF_OK;
    }
};
}  // namespace

// staging of the exchange's status words for a communicator of `world` ranks (allreduce_over): G sizes + status per rank, G ranks
static int comm_words_ensure(vf_ctx *c, int world) {
    const size_t W1 = (size_t)world + 1;
    return c->d_comm_words.ensure(W1 + W1 * (size_t)world + 16);
}
static void vf_comm_release(vf_ctx *c) {
    if (c->comm) { if (Rccl *R = rccl()) (void)R->CommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
    c->local_group = nullptr;
    c->comm_rank = 0; c->comm_world = 1;
}

static int vf_comm_unique_id_impl(void *id128);
extern "C" int vf_comm_unique_id(void *id128) { return vf_guard([&] { return vf_comm_unique_id_impl(id128); }); }
static int vf_comm_unique_id_impl(void *id128) {
    if (!id128) return fail(VF_EINVAL, "null argument");
    Rccl *R = rccl();
    if (!R) return fail(VF_EDEVICE, "librccl.so.1 is not loadable");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(R->GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return VF_OK;
}

static int vf_comm_init_impl(vf_ctx *c, int rank, int world, const void *id128);
extern "C" int vf_comm_init(vf_ctx *c, int rank, int world, const void *id128) { return vf_guard([&] { return vf_comm_init_impl(c, rank, world, id128); }); }
static int vf_comm_init_impl(vf_ctx *c, int rank, int world, const void *id128) {
    if (!c || !id128) return fail(VF_EINVAL, "null argument");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(VF_EINVAL, "rank / world out of range (1..64 GPUs)");
    Rccl *R = rccl();
    if (!R) return fail(VF_EDEVICE, "librccl.so.1 is not loadable");
    HIP_TRY(hipSetDevice(c->dev));
    vf_comm_release(c);
    { const int wrc = comm_words_ensure(c, world); if (wrc) return wrc; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    NCCL_TRY(R->CommInitRank(&comm, world, id, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    return VF_OK;
}

static int vf_comm_local_create_impl(int world, vf_local_group **out);
extern "C" int vf_comm_local_create(int world, vf_local_group **out) { return vf_guard([&] { return vf_comm_local_create_impl(world, out); }); }
static int vf_comm_local_create_impl(int world, vf_local_group **out) {
    if (!out) return fail(VF_EINVAL, "null argument");
    *out = nullptr;
    if (world < 1 || world > 64) return fail(VF_EINVAL, "world out of range (1..64 contexts)");
    std::unique_ptr<vf_local_group> g(new vf_local_group);
    g->world = world;
    g->gather.resize((size_t)world);
    g->mail.resize((size_t)world * (size_t)world);
    *out = g.release();
    return VF_OK;
}
extern "C" void vf_comm_local_destroy(vf_local_group *g) { delete g; }

static int vf_comm_init_local_impl(vf_ctx *c, int rank, vf_local_group *g);
extern "C" int vf_comm_init_local(vf_ctx *c, int rank, vf_local_group *g) { return vf_guard([&] { return vf_comm_init_local_impl(c, rank, g); }); }
static int vf_comm_init_local_impl(vf_ctx *c, int rank, vf_local_group *g) {
    if (!c || !g) return fail(VF_EINVAL, "null argument");
    if (rank < 0 || rank >= g->world) return fail(VF_EINVAL, "rank out of range for this group");
    vf_comm_release(c);
    HIP_TRY(hipSetDevice(c->dev));
    { const int wrc = comm_words_ensure(c, g->world); if (wrc) return wrc; }
    c->local_group = g; c->comm_rank = rank; c->comm_world = g->world;
    return VF_OK;
}

// Owner-partitioned reduce of the count tables of all ranks (primary design; the multi-GPU form of the serial
// reduce of lib.rs:292-306):
//   1. every rank buckets its entries by owner (hash tag -> one of `world`); sizes and a status word go round (all-gather 1);
//   2. every rank sizes and allocates ALL its buffers from those sizes - receive side of the bucket exchange, its share,
//      root's receive side of the share gather (upper bounds: merged blobs only shrink) - and the statuses go round again
//      (all-gather 2), so that nobody enters the exchange unless everybody can;
//   3. ONE round of sends / receives: bucket o of every rank to rank o (each entry crosses xGMI once, all links of a GPU
//      busy at once, no ring);
//   4. every owner empties its table, merges its own bucket and the world - 1 it received - its share, fully summed - and
//      exports it; share sizes and statuses go round (all-gather 3);
//   5. the shares - disjoint key sets - go to `root` in one round, which merges them.
// A rank that fails locally (no memory for a buffer, a table that cannot grow) says so in the next status word and every
// rank leaves with an error together: nobody is left waiting in a receive for a peer that has gone (round 2 returned early
// between the collectives).  Afterwards root's table is the sum of all tables; every other rank holds its own share.
static int allreduce_over(vf_ctx *c, Transport &T, int root) {
    const int G = c->comm_world, me = c->comm_rank;
    hipStream_t st = c->main_stream;
    DevBuf<uint8_t> sendbuf, recvbuf, sharebuf, gatherbuf;
    struct Free { DevBuf<uint8_t> &a, &b, &c2, &d; ~Free() { a.release(); b.release(); c2.release(); d.release(); } } fr{sendbuf, recvbuf, sharebuf, gatherbuf};
    const size_t W1 = (size_t)G + 1;                                   // words a rank contributes to all-gather 1: G sizes + status
    // The staging of the status words was allocated when the communicator was made (comm_words_ensure): the one thing the
    // "every rank leaves together" promise cannot survive is a rank that cannot SAY it has failed.
    DevBuf<uint64_t> &d_words = c->d_comm_words;
    if (d_words.cap < W1 + W1 * (size_t)G) return fail(VF_EINVAL, "table exchange: the communicator has no staging words (vf_comm_init / vf_comm_init_local)");
    // statuses of all ranks after a gather of `words` words per rank whose last word is the status; VF_OK only if all are
    auto agreed = [&](const std::vector<uint64_t> &all, size_t words, int mine) -> int {
        if (mine) return mine;                                         // (own message already set)
        for (int r = 0; r < G; ++r)
            if (all[(size_t)r * words + words - 1] != 0)
                return fail((int)all[(size_t)r * words + words - 1], "table exchange: rank " + std::to_string(r) + " failed (" + std::to_string(all[(size_t)r * words + words - 1]) + "), every rank leaves");
        return VF_OK;
    };
    // a transport call failed, or a wait gave up: close what is open, break the communicator (no peer stays blocked on this rank)
    auto broken = [&](int code) -> int { T.abort(); return code; };
    auto gather_words = [&](const std::vector<uint64_t> &mine_w, std::vector<uint64_t> &all_w) -> int {
        const size_t words = mine_w.size();
        all_w.assign(words * (size_t)G, 0);
        if (hipMemcpyAsync(d_words.p, mine_w.data(), words * sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess) return broken(fail(VF_EDEVICE, "hipMemcpyAsync failed"));
        int r2 = T.all_gather(d_words.p, d_words.p + words, words, st);
        if (r2) return broken(r2);
        if (hipMemcpyAsync(all_w.data(), d_words.p + words, words * (size_t)G * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess) return broken(fail(VF_EDEVICE, "hipMemcpyAsync failed"));
        if ((r2 = T.wait(st))) return broken(r2);
        return VF_OK;
    };
    // one round: the bracket is closed on every way out (an error between ncclGroupStart and ncclGroupEnd used to return with the
    // group open and the peers inside their ncclGroupEnd)
    auto round = [&](auto &&body) -> int {
        int r2 = T.begin();
        if (r2) return broken(r2);
        r2 = body();
        if (r2) return broken(r2);                                     // (abort() closes the open round first)
        if ((r2 = T.end(st))) return broken(r2);
        if ((r2 = T.wait(st))) return broken(r2);
        return VF_OK;
    };
    int rc = VF_OK;

    // 1. buckets
    std::vector<uint64_t> mine(G, 0), soff(G + 1, 0);
    if (!rc && allreduce_test_fail(me, 1)) rc = fail(VF_ENOMEM, "injected failure (VFIND_TEST_FAIL_ALLREDUCE)");
    if (!rc) rc = vf_table_partition_sizes(c, (uint32_t)G, mine.data());
    for (int o = 0; o < G; ++o) soff[o + 1] = soff[o] + ((mine[o] + 255) & ~255ull);
    if (!rc) rc = sendbuf.ensure(soff[G] + 256);
    for (int o = 0; o < G && !rc; ++o) rc = vf_table_partition_export(c, (uint32_t)G, (uint32_t)o, sendbuf.p + soff[o], mine[o], nullptr);
    std::vector<uint64_t> w1(mine), all1;
    w1.push_back((uint64_t)rc);
    int grc = gather_words(w1, all1);
    if (grc) return grc;
    if ((rc = agreed(all1, W1, rc))) return rc;
    auto bucket = [&](int from, int owner) { return all1[(size_t)from * W1 + owner]; };

    // 2. every buffer this rank will need, from upper bounds
    std::vector<uint64_t> roff(G + 1, 0);
    for (int q = 0; q < G; ++q) roff[q + 1] = roff[q] + ((bucket(q, me) + 255) & ~255ull);
    uint64_t share_cap = 0;                                            // my share: the sum of what it is merged from
    for (int q = 0; q < G; ++q) share_cap += bucket(q, me);
    std::vector<uint64_t> gcap(G + 1, 0);                              // root: room for every other rank's share
    if (me == root)
        for (int q = 0; q < G; ++q) {
            uint64_t cap_q = 0;
            for (int p = 0; p < G; ++p) cap_q += bucket(p, q);
            gcap[q + 1] = gcap[q] + (q == me ? 0 : ((cap_q + 255) & ~255ull));
        }
    if (allreduce_test_fail(me, 2)) rc = fail(VF_ENOMEM, "injected failure (VFIND_TEST_FAIL_ALLREDUCE)");
    if (!rc) rc = recvbuf.ensure(roff[G] + 256);
    if (!rc && me != root) rc = sharebuf.ensure(share_cap + 256);
    if (!rc && me == root) rc = gatherbuf.ensure(gcap[G] + 256);
    std::vector<uint64_t> w2{(uint64_t)rc}, all2;
    if ((grc = gather_words(w2, all2))) return grc;
    if ((rc = agreed(all2, 1, rc))) return rc;

    // 3. bucket `me` of every rank comes here
    if ((rc = round([&]() -> int {
            int r2 = VF_OK;
            for (int p = 0; p < G && !r2; ++p) {
                if (p == me) continue;
                r2 = T.send(sendbuf.p + soff[p], mine[p], p, st);
                if (!r2) r2 = T.recv(recvbuf.p + roff[p], bucket(p, me), p, st);
            }
            return r2;
        }))) return rc;                                                // (the transport itself failed: communicator aborted, nothing to agree over)

    // 4. this rank's share
    if (allreduce_test_fail(me, 3)) rc = fail(VF_ENOMEM, "injected failure (VFIND_TEST_FAIL_ALLREDUCE)");
    if (!rc) rc = vf_table_clear(c);
    if (!rc) rc = vf_table_merge(c, sendbuf.p + soff[me], mine[me]);
    for (int q = 0; q < G && !rc; ++q)
        if (q != me) rc = vf_table_merge(c, recvbuf.p + roff[q], bucket(q, me));
    uint64_t share = 0;
    if (!rc) rc = vf_table_export_size(c, &share);
    if (!rc && me != root) {
        if (share > share_cap) rc = fail(VF_EDEVICE, "table exchange: a merged share is larger than its parts (internal error)");
        else rc = vf_table_export(c, sharebuf.p, share, nullptr);
    }
    std::vector<uint64_t> w3{share, (uint64_t)rc}, all3;
    if ((grc = gather_words(w3, all3))) return grc;
    if ((rc = agreed(all3, 2, rc))) return rc;

    // 5. shares to root
    if ((rc = round([&]() -> int {
            if (me != root) return T.send(sharebuf.p, share, root, st);
            int r2 = VF_OK;
            for (int q = 0; q < G && !r2; ++q) {
                if (q == me) continue;
                if (all3[(size_t)q * 2] > gcap[q + 1] - gcap[q]) r2 = fail(VF_EDEVICE, "table exchange: a share is larger than its parts (internal error)");
                else r2 = T.recv(gatherbuf.p + gcap[q], all3[(size_t)q * 2], q, st);
            }
            return r2;
        }))) return rc;
    if (me == root)
        for (int q = 0; q < G; ++q)
            if (q != me && (rc = vf_table_merge(c, gatherbuf.p + gcap[q], all3[(size_t)q * 2]))) return rc;
    return VF_OK;
}

static int vf_table_allreduce_impl(vf_ctx *c, int root);
extern "C" int vf_table_allreduce(vf_ctx *c, int root) { return vf_guard([&] { return vf_table_allreduce_impl(c, root); }); }
static int vf_table_allreduce_impl(vf_ctx *c, int root) {
    if (!c) return fail(VF_EINVAL, "null context");
    const int G = c->comm_world;
    // (test hook VFIND_TEST_ALLREDUCE_SINGLE: a one-rank communicator walks the whole exchange - every transport call, wait
    // and round with nobody to talk to - which is as far as RcclTransport can be run on a one-GPU box)
    if (G == 1 && !(getenv("VFIND_TEST_ALLREDUCE_SINGLE") && (c->comm || c->local_group))) return VF_OK;
    if (!c->comm && !c->local_group) return fail(VF_EINVAL, "vf_comm_init has not been called");
    if (root < 0 || root >= G) return fail(VF_EINVAL, "root out of range");
    HIP_TRY(hipSetDevice(c->dev));
    if (c->local_group) {
        LocalTransport T(c->local_group, c->comm_rank, c->dev);
        const int rc = allreduce_over(c, T, root);
        if (rc) T.abort();                                   // peers waiting at the rendezvous leave too (the message is kept)
        return rc;
    }
    Rccl *R = rccl();
    if (!R) return fail(VF_EDEVICE, "librccl.so.1 is not loadable");
    RcclTransport T(R, (ncclComm_t)c->comm, &c->comm);
    // A rank that leaves with an error of its OWN between two collectives has said so in a status word and its peers leave
    // with it: the communicator stays usable.  A transport error or a deadline has aborted it inside allreduce_over.
    return allreduce_over(c, T, root);
}

static int vf_finish_impl(vf_ctx *c, vf_result **out);
extern "C" int vf_finish(vf_ctx *c, vf_result **out) { return vf_guard([&] { return vf_finish_impl(c, out); }); }
static int vf_finish_impl(vf_ctx *c, vf_result **out) {
    if (vf_internal_test_throw("finish")) throw std::bad_alloc();
    if (vf_internal_test_throw("finish_length")) { std::vector<uint64_t> v; v.resize((size_t)-1 / 2); }     // std::length_error
    if (!c || !out) return fail(VF_EINVAL, "null argument");
    *out = nullptr;
    uint64_t nbytes = 0;
    int rc = vf_table_export_size(c, &nbytes);
    if (rc) return rc;
    DevBuf<uint8_t> blob;
    if ((rc = blob.ensure(nbytes))) return rc;
    struct Free { DevBuf<uint8_t> &b; ~Free() { b.release(); } } fr{blob};
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = vf_table_export(c, blob.p, nbytes, nullptr))) return rc;
    // the row set of the DataFrame (lib.rs:312-316) is built on the device (vf_result.hip): strings expanded, duplicate
    // strings summed (possible only if the device saw a stale table word, see vf_table_dev.h:table_add), rows ordered by
    // count descending then sequence ascending; only the three finished buffers come back
    uint64_t hdr[2];
    HIP_TRY(hipMemcpy(hdr, blob.p, 16, hipMemcpyDeviceToHost));
    const uint64_t n = hdr[0];
    std::unique_ptr<vf_result> r(new vf_result);
    std::string err;
    rc = vf::build_result_on_device(c->main_stream, (const uint64_t *)(blob.p + 16), (const uint64_t *)(blob.p + 16 + n * 8),
                                    (const uint32_t *)(blob.p + 16 + n * 16), n, c->skip_translation, *r, err);
    if (rc) return fail(rc, err);
    c->ms_result = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->result_rows = r->counts.size();
    c->dup_rows = r->dup_rows;
    *out = r.release();
    return VF_OK;
}

// The frame of several exported tables at once (the tables of several GPUs gathered on one device, or the shares of an
// owner-partitioned exchange): laid end to end and handed to the device result build, which sums equal strings -- no merge
// into a count table in between.
static int vf_finish_blobs_impl(vf_ctx *c, const void *const *d_blobs, const uint64_t *nbytes, uint32_t n_blobs, vf_result **out);
extern "C" int vf_finish_blobs(vf_ctx *c, const void *const *d_blobs, const uint64_t *nbytes, uint32_t n_blobs, vf_result **out) { return vf_guard([&] { return vf_finish_blobs_impl(c, d_blobs, nbytes, n_blobs, out); }); }
static int vf_finish_blobs_impl(vf_ctx *c, const void *const *d_blobs, const uint64_t *nbytes, uint32_t n_blobs, vf_result **out) {
    if (!c || !out || (n_blobs && (!d_blobs || !nbytes))) return fail(VF_EINVAL, "null argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(c->dev));
    int rc;
    std::vector<uint64_t> n(n_blobs), adw(n_blobs);
    uint64_t tot_n = 0, tot_adw = 0;
    // the headers in stream order, behind whatever vf_wait_stream queued on the main stream (read_blob_header)
    if ((rc = c->h_hdr.ensure(2 * (size_t)n_blobs + 2))) return rc;
    for (uint32_t k = 0; k < n_blobs; ++k) {
        if (!d_blobs[k] || nbytes[k] < 16) return fail(VF_EINVAL, "blob too small");
        HIP_TRY(launch_peek(c->main_stream, d_blobs[k], c->h_hdr.p + 2 * (size_t)k, 4));
    }
    HIP_TRY(hipStreamSynchronize(c->main_stream));
    for (uint32_t k = 0; k < n_blobs; ++k) {
        const uint64_t hdr[2] = {c->h_hdr.p[2 * (size_t)k], c->h_hdr.p[2 * (size_t)k + 1]};
        if ((rc = blob_header_ok(hdr, nbytes[k]))) return rc;
        n[k] = hdr[0]; adw[k] = hdr[1];
        tot_n += hdr[0]; tot_adw += hdr[1];
    }
    if (tot_adw >> 39) return fail(VF_ECAPACITY, "gathered tables hold more key dwords than an entry reference can address");
    const auto t0 = std::chrono::steady_clock::now();
    DevBuf<uint64_t> refs, counts;
    DevBuf<uint32_t> arena;
    struct Free { DevBuf<uint64_t> &a, &b; DevBuf<uint32_t> &c; ~Free() { a.release(); b.release(); c.release(); } } fr{refs, counts, arena};
    if ((rc = refs.ensure(tot_n)) || (rc = counts.ensure(tot_n)) || (rc = arena.ensure(tot_adw))) return rc;
    uint64_t at_n = 0, at_adw = 0;
    for (uint32_t k = 0; k < n_blobs; ++k) {
        const uint8_t *b = (const uint8_t *)d_blobs[k];
        HIP_TRY(vf::rebase_refs(c->main_stream, (const uint64_t *)(b + 16), n[k], at_adw, refs.p + at_n));
        if (n[k]) HIP_TRY(hipMemcpyAsync(counts.p + at_n, b + 16 + n[k] * 8, n[k] * 8, hipMemcpyDeviceToDevice, c->main_stream));
        if (adw[k]) HIP_TRY(hipMemcpyAsync(arena.p + at_adw, b + 16 + n[k] * 16, adw[k] * 4, hipMemcpyDeviceToDevice, c->main_stream));
        at_n += n[k]; at_adw += adw[k];
    }
    std::unique_ptr<vf_result> r(new vf_result);
    std::string err;
    rc = vf::build_result_on_device(c->main_stream, refs.p, counts.p, arena.p, tot_n, c->skip_translation, *r, err);
    if (rc) return fail(rc, err);
    c->ms_result = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->result_rows = r->counts.size();
    c->dup_rows = r->dup_rows;
    *out = r.release();
    return VF_OK;
}

extern "C" uint64_t vf_result_rows(const vf_result *r) { return r ? r->counts.size() : 0; }
extern "C" const uint8_t *vf_result_key_bytes(const vf_result *r) { return r ? r->keys.data() : nullptr; }
extern "C" const uint64_t *vf_result_key_offsets(const vf_result *r) { return r ? r->offsets.data() : nullptr; }
extern "C" const uint64_t *vf_result_counts(const vf_result *r) { return r ? r->counts.data() : nullptr; }
extern "C" void vf_result_free(vf_result *r) { delete r; }

// ------------------------------------------------------------------------------------------------
// stats
// ------------------------------------------------------------------------------------------------
static int vf_stats_get_impl(vf_ctx *c, vf_stats *o);
extern "C" int vf_stats_get(vf_ctx *c, vf_stats *o) { return vf_guard([&] { return vf_stats_get_impl(c, o); }); }
static int vf_stats_get_impl(vf_ctx *c, vf_stats *o) {
    if (!c || !o) return fail(VF_EINVAL, "null argument");
    int rc = vf_sync(c);
    if (rc && rc != VF_ECAPACITY) return rc;
    uint64_t h[C_NUM];
    HIP_TRY(hipMemcpy(h, c->t_counters.p, sizeof(h), hipMemcpyDeviceToHost));
    if (c->spare.valid) {                                  // read statistics outlive a clear: they sit in both counter arrays
        uint64_t h2[C_NUM];
        HIP_TRY(hipMemcpy(h2, c->spare.counters.p, sizeof(h2), hipMemcpyDeviceToHost));
        for (int k = kTableCounters; k < C_NUM; ++k) h[k] += h2[k];
    }
    memset(o, 0, sizeof(*o));
    o->reads = c->n_reads; o->reads_generic = c->n_generic;
    o->dp_prefix = h[C_DP_PREFIX]; o->dp_suffix = h[C_DP_SUFFIX];
    o->accepted = h[C_ACCEPTED]; o->counted = h[C_COUNTED]; o->insert_path = h[C_SLOW];
    o->table_entries = h[C_ENTRIES]; o->arena_dwords = h[C_KEY_DWORDS]; o->arena_top = h[C_ARENA_TOP];
    o->ms_scan = c->ms[K_SCAN]; o->ms_filter = c->ms[K_FILTER]; o->ms_dp_window = c->ms[K_DPW]; o->ms_dp_rest = c->ms[K_DPREST];
    o->ms_finish = c->ms[K_FIN]; o->ms_generic = c->ms[K_GEN];
    o->ms_result_build = c->ms_result; o->result_rows = c->result_rows;
    for (int k = 0; k < 4; ++k) o->dbg[k] = h[C_DBG0 + k];
    o->launches_scan = c->launches[K_SCAN]; o->launches_filter = c->launches[K_FILTER];
    o->launches_dp_window = c->launches[K_DPW]; o->launches_dp_rest = c->launches[K_DPREST];
    o->launches_finish = c->launches[K_FIN]; o->launches_generic = c->launches[K_GEN];
    o->ms_count_log = c->ms[K_LOG]; o->launches_count_log = c->launches[K_LOG];
    o->dup_rows = c->dup_rows;
    const VfFastqStats &f = c->fq;
    o->fq_wall_ms = f.wall_ns.load() * 1e-6; o->fq_inflate_ms = f.inflate_ns.load() * 1e-6; o->fq_seq_wait_ms = f.seq_wait_ns.load() * 1e-6;
    o->fq_queue_full_ms = f.queue_full_ns.load() * 1e-6; o->fq_submit_ms = f.submit_ns.load() * 1e-6; o->fq_slot_wait_ms = f.slot_wait_ns.load() * 1e-6;
    o->fq_index_wait_ms = f.index_wait_ns.load() * 1e-6; o->fq_drain_ms = f.drain_ns.load() * 1e-6;
    o->fq_comp_bytes = f.comp_bytes.load(); o->fq_text_bytes = f.text_bytes.load(); o->fq_inflate_threads = f.threads.load(); o->fq_chunks = f.chunks.load();
    o->ms_h2d = c->ms[K_H2D]; o->launches_h2d = c->launches[K_H2D]; o->ms_frame = c->ms[K_FRAME]; o->launches_frame = c->launches[K_FRAME];
    return VF_OK;
}
static int vf_stats_reset_impl(vf_ctx *c);
extern "C" int vf_stats_reset(vf_ctx *c) { return vf_guard([&] { return vf_stats_reset_impl(c); }); }
static int vf_stats_reset_impl(vf_ctx *c) {
    if (!c) return fail(VF_EINVAL, "null context");
    int rc = vf_sync(c);
    if (rc && rc != VF_ECAPACITY) return rc;
    c->n_reads = 0; c->n_generic = 0;
    for (int k = 0; k < K_KINDS; ++k) { c->ms[k] = 0; c->launches[k] = 0; }
    c->fq.reset();
    const uint64_t z[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(c->t_counters.p + C_ACCEPTED, z, 4 * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->t_counters.p + C_SLOW, z, sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->t_counters.p + C_DBG0, z, 4 * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (c->spare.valid) {
        HIP_TRY(hipMemcpy(c->spare.counters.p + C_ACCEPTED, z, 4 * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->spare.counters.p + C_SLOW, z, sizeof(uint64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->spare.counters.p + C_DBG0, z, 4 * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    return VF_OK;
}
static int vf_set_profiling_impl(vf_ctx *c, int on);
extern "C" int vf_set_profiling(vf_ctx *c, int on) { return vf_guard([&] { return vf_set_profiling_impl(c, on); }); }
static int vf_set_profiling_impl(vf_ctx *c, int on) {
    if (!c) return fail(VF_EINVAL, "null context");
    c->profiling = (on == 2 || on == 3) ? on : (on != 0 ? 1 : 0);
    c->prof_batch = 0;
    return VF_OK;
}

// ------------------------------------------------------------------------------------------------
// hooks for the FASTQ reader (vf_fastq.cpp)
// ------------------------------------------------------------------------------------------------
// A chunk of whole FASTQ records (byte ranges of inflated text): copied into pinned staging, sent to the
// device, indexed by line, framed and packed there (k_nl_*, k_frame_pack), then the usual batch path.
VfFastqStats *vf_internal_fastq_stats(vf_ctx *c) { return &c->fq; }
bool vf_internal_device_framing(const vf_ctx *) {
    static const bool on = [] { const char *e = getenv("VFIND_FASTQ_DEVICE_FRAMING"); return !e || atoi(e) != 0; }();
    return on;
}
uint8_t *vf_internal_alloc_pinned(const vf_ctx *c, size_t bytes) {
    void *p = nullptr;
    if (hipSetDevice(c->dev) != hipSuccess) return nullptr;      // called from reader threads: stay on the context's device
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;      // any device of the process may read it
    return (uint8_t *)p;
}
void vf_internal_free_pinned(uint8_t *p) { if (p) (void)hipHostFree(p); }

int vf_internal_submit_text(vf_ctx *c, const TextSpan *spans, size_t n_spans, bool final_chunk, bool spans_pinned) {
    size_t total = 0;
    for (size_t i = 0; i < n_spans; ++i) total += spans[i].n;
    if (total == 0) return VF_OK;
    if (total > 0x7FFFFFF0ull) return fail(VF_EINVAL, "text chunk too large");
    HIP_TRY(hipSetDevice(c->dev));
    const auto t_in = std::chrono::steady_clock::now();
    auto ns_since = [](std::chrono::steady_clock::time_point t0) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); };
    struct Whole { vf_ctx *c; std::chrono::steady_clock::time_point t0; ~Whole() { c->fq.submit_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); c->fq.chunks += 1; } } whole{c, t_in};
    Slot *s = acquire_slot(c);
    struct Rel { vf_ctx *c; Slot *s; ~Rel() { release_slot(c, s); } } rel{c, s};
    int rc;
    if ((rc = wait_slot(c, s))) return rc;
    c->fq.slot_wait_ns += ns_since(t_in);
    const int pmode = c->profiling.load();
    // final chunk: trailing newlines are padding, and the last record may lack its newline
    size_t n_use = n_spans;
    size_t last_n = n_spans ? spans[n_spans - 1].n : 0;
    bool add_nl = false;
    if (final_chunk) {
        while (n_use) {
            const uint8_t *p = spans[n_use - 1].p;
            while (last_n && (p[last_n - 1] == '\n' || p[last_n - 1] == '\r')) { --last_n; --total; }
            if (last_n) break;
            --n_use;
            last_n = n_use ? spans[n_use - 1].n : 0;
        }
        if (total == 0) return VF_OK;
        add_nl = true;
        ++total;
    }
    const size_t tiles = (total + 4095) / 4096;
    if ((rc = s->d_text.ensure(total + 64))) return rc;
    if ((rc = s->d_tile_cnt.ensure(tiles))) return rc;
    if ((rc = s->d_tile_off.ensure(tiles))) return rc;
    if ((rc = s->d_line_end.ensure(total + 1))) return rc;
    if ((rc = s->d_info.ensure(4))) return rc;
    if ((rc = s->h_info.ensure(4))) return rc;
    {
    Timed th2d(c, s->stream, K_H2D, pmode, false);
    if (spans_pinned) {
        // the pieces are page-locked: the DMA engine reads them directly (they stay referenced until this call returns,
        // and the stream is waited for below)
        size_t at = 0;
        for (size_t i = 0; i < n_use; ++i) {
            const size_t n = i + 1 == n_use ? last_n : spans[i].n;
            if (n) HIP_TRY(hipMemcpyAsync(s->d_text.p + at, spans[i].p, n, hipMemcpyHostToDevice, s->stream));
            at += n;
        }
        if (add_nl) { static const uint8_t nl = '\n'; HIP_TRY(hipMemcpyAsync(s->d_text.p + at, &nl, 1, hipMemcpyHostToDevice, s->stream)); }
    } else {
        if ((rc = s->h_text.ensure(total + 64))) return rc;
        size_t at = 0;
        for (size_t i = 0; i < n_use; ++i) {
            const size_t n = i + 1 == n_use ? last_n : spans[i].n;
            memcpy(s->h_text.p + at, spans[i].p, n);
            at += n;
        }
        if (add_nl) s->h_text.p[at] = '\n';
        HIP_TRY(hipMemcpyAsync(s->d_text.p, s->h_text.p, total, hipMemcpyHostToDevice, s->stream));
    }
    }
    { Timed tf(c, s->stream, K_FRAME, pmode, false); HIP_TRY(launch_line_index(s->stream, s->d_text.p, total, s->d_tile_cnt.p, s->d_tile_off.p, s->d_line_end.p, s->d_info.p)); }
    HIP_TRY(launch_peek(s->stream, s->d_info.p, s->h_info.p, 2));
    { const auto tw = std::chrono::steady_clock::now(); HIP_TRY(hipStreamSynchronize(s->stream)); c->fq.index_wait_ns += ns_since(tw); }
    const uint32_t n_lines = s->h_info.p[0], max_line = s->h_info.p[1];
    if (n_lines % 4u) return fail(VF_EFORMAT, "malformed FASTQ: truncated record");
    const uint64_t n = n_lines / 4u;
    if (n == 0) return VF_OK;
    CapacityTicket ticket{c};
    if ((rc = reserve_capacity(c, ticket, n, n * arena_need(c, max_line)))) return rc;
    std::shared_lock<std::shared_mutex> table_lock(c->table_mu);
    uint32_t W = c->fast_ok ? vf_row_dwords_for(std::min<uint32_t>(std::max<uint32_t>(max_line, 1u), (uint32_t)kMaxFastRead)) : 10u;
    if (W == 0) W = 40;
    if ((rc = s->d_rows.ensure((size_t)n * W))) return rc;
    if ((rc = s->d_lens.ensure(n))) return rc;
    if ((rc = s->ws.ensure(n, (int)W))) return rc;
    if ((rc = s->d_gen_start.ensure(n))) return rc;
    if ((rc = s->d_gen_len.ensure(n))) return rc;
    // adapters / scores outside the 2-bit path: every read goes the byte-wise way (frame_pack lists reads longer than 0 rows)
    { Timed tf(c, s->stream, K_FRAME, pmode, false);
      HIP_TRY(launch_frame_pack(s->stream, c->fast_ok ? (int)W : 10, s->d_text.p, s->d_line_end.p, (uint32_t)n, s->d_rows.p, s->d_lens.p,
                                s->d_gen_start.p, s->d_gen_len.p, s->d_info.p, !c->fast_ok)); }
    c->n_reads += n;
    if (c->fast_ok) {
        if ((rc = run_packed(c, s->stream, s->index, s->ws, s->d_rows.p, (int)W, s->d_lens.p, n, s->ws.spans.p, true, s->h_entries.p))) return rc;
    }
    {   // listed reads (non-ACGT bytes, longer than a row): byte-wise kernel over the same text, count known only on the device
        const uint32_t mmax = (uint32_t)std::max(c->pre.size(), c->suf.size());
        const uint32_t rows = 4 * mmax + (max_line + 3) / 4 + 4;
        const uint64_t threads = 16384;
        if ((rc = s->d_gws.ensure((size_t)rows * threads))) return rc;
        GenericParams gp = c->gp;
        gp.count_mode = 1;
        Table tab = c->tab;
        tab.chunk_row = s->index;
        HIP_TRY(hipStreamWaitEvent(s->stream, c->clear_event, 0));
        Timed t(c, s->stream, K_GEN, c->profiling.load(), false);
        HIP_TRY(launch_generic_indexed(s->stream, s->d_text.p, s->d_gen_start.p, s->d_gen_len.p, s->d_info.p + 3, gp, s->d_gws.p, threads, rows, tab));
    }
    HIP_TRY(launch_peek(s->stream, s->d_info.p, s->h_info.p, 4));
    HIP_TRY(hipEventRecord(s->done, s->stream));
    capacity_note_enqueued(c, s, n);
    s->text_pending = true;
    return VF_OK;
}

int vf_internal_submit(vf_ctx *c, const uint8_t *base, const ReadRef *reads, uint64_t n) {
    int rc = submit_reads(c, base, reads, n, nullptr, nullptr);
    return rc;
}
void vf_internal_set_async_error(vf_ctx *c, int rc, const std::string &msg) {
    std::lock_guard<std::mutex> g(c->err_mu);
    if (!c->async_rc) { c->async_rc = rc; c->async_msg = msg; }
}
uint32_t vf_internal_threads(const vf_ctx *c) { return c->n_threads; }
int vf_internal_fail(int code, const std::string &msg) { return fail(code, msg); }
std::string vf_internal_last_error() { return g_err; }
