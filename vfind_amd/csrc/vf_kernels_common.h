// vf_kernels_common.h -- what the kernel translation units of the hot path share (device helpers, the work-list
// conventions, launch helpers).  gfx950 (MI355X / CDNA4): one read (or task) per lane, 64-lane waves, integer VALU only (no
// MFMA: there is no dense contraction in this path).  The kernels, by translation unit (reference: vfind 0.3.1, src/lib.rs):
//   vf_k_front.hip   k_scan      memmem::find for both adapters (lib.rs:148) + the control flow of find_adapter_match
//                                (lib.rs:141-166): eight pattern bases at every start position, the whole pattern at the survivors;
//                                appends reads that need an alignment to per-workgroup lists
//                    k_filter / k_bmfilter   exact pre-filters of those reads: block filter, Myers bit-vector edit distance of the
//                                adapter against every column, Hamming certificate -> task classes
//   vf_k_dp.hip      k_dpall     parasail sg_stats_scan semi-global affine DP with the `length` statistic (lib.rs:155-160), tie policy P
//                                of SURVEY.md App. B.3, for every task class of the filter in one launch: over a band of 5 / 10 / 16
//                                diagonals, over the window(s), over the whole read; k_dp = the whole-read DP alone when no filter applies
//   vf_k_table.hip   k_finish    seq[start..end] extraction (lib.rs:288-290), translate (lib.rs:16-44) and the HashMap<String,u64>
//                                reduce (lib.rs:292-306) as a device hash-count table (vf_table_dev.h); the count log; k_generic = the
//                                same per-read function byte-wise, for reads / adapters the 2-bit fast path cannot hold; k_export /
//                                k_merge / k_rehash = table compaction, (key,count) merge for the multi-GPU exchange, growth
//   vf_k_text.hip    k_nl_* / k_frame_pack   FASTQ text on the device: newline index, four-line framing, 2-bit packing
#pragma once
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

// ---- task classes shared by the filters (which assign them) and the DP (which runs them) ----
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
constexpr int kHamDiags = 12;     // diagonals the Hamming certificate lays the adapter on (hamming_band_class)
constexpr int kBandBias = 4096;  // class-0 window word = first diagonal + kBandBias
#ifndef VF_HAMMING_CERT
#define VF_HAMMING_CERT 1       // (A/B: make variant NAME=noham DEFS=-DVF_HAMMING_CERT=0)
#endif

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
static inline int grid_for(uint64_t n, int per_cu) {
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
static inline int grid_fit_mask() {
    static const int m = [] { const char *v = getenv("VFIND_GRID_BY_OCCUPANCY"); return v ? atoi(v) : 3; }();
    return m;
}
template <class K> static inline int occupancy_of(K kernel, size_t dyn_lds) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlock, dyn_lds) != hipSuccess) { (void)hipGetLastError(); nb = 1; }
    return nb < 1 ? 1 : nb;
}
static inline int w_index(int W) { return W == 10 ? 0 : W == 20 ? 1 : 2; }

}  // namespace vf
