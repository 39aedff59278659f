// vf_k_dp.hip -- the semi-global affine DP with the length statistic: k_dpall (every task class of a filter pass), k_dp (whole reads)
// (kernels of the variant-extraction hot path, gfx950; the map of the translation units is in vf_kernels_common.h)
#include "vf_kernels_common.h"

namespace vf {

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
// launchers
// ------------------------------------------------------------------------------------------------
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


#ifdef VF_FINISH_ABLATION
extern "C" int vf_debug_dp_trace(uint64_t *out) {          // diagnosis build: 2 x kMaxGrid x 4 words (scripts/dp_trace.py)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dp_trace), sizeof(uint64_t) * 2 * kMaxGrid * 4) == hipSuccess ? 0 : 4;
}
#endif

}  // namespace vf
