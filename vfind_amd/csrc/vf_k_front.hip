// vf_k_front.hip -- k_scan (exact search, spans, task lists) and the pre-DP filters k_filter / k_bmfilter
// (kernels of the variant-extraction hot path, gfx950; the map of the translation units is in vf_kernels_common.h)
#include "vf_kernels_common.h"

namespace vf {

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
// (one v_bitop3 per accumulated base: (x ^ b) | acc.  Left to itself the optimiser pairs the terms of the straight-line form
// into two XORs and a v_or3 - three instructions for two bases)
#define VF_KEEP(x) asm volatile("" : "+v"(x))
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

// P44: both adapters are 20 bases or longer, i.e. both plans are {4, 4} (four pairs, no singles): the stage-1 loop is then
// straight-line code.  With the plans read at run time it is sixteen wave-uniform branches a trip whose conditions the
// compiler keeps as 64-bit masks - more than the SGPR file holds, so ~70 v_readlane a trip fetch them back from a VGPR.
template <int W, bool P44>
__device__ __forceinline__ void exact_find2(const uint32_t (&r)[W + 2], const uint32_t *row, int n, const ScanParams &sp,
                                            int &ps, int &ss) {
    uint32_t accP[W], accS[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { accP[w] = 0; accS[w] = 0; }
    const int mp = (int)sp.pre_len, ms = (int)sp.suf_len;
    const ScanPlan pp = P44 ? ScanPlan{4, 4} : scan_plan(mp), sq = P44 ? ScanPlan{4, 4} : scan_plan(ms);
#pragma unroll
    for (int b = 0; b < (P44 ? 4 : 16); ++b) {
        const bool p0 = b < pp.nlo, p1 = b < pp.nhi, s0 = b < sq.nlo, s1 = b < sq.nhi;       // wave-uniform
        if (p0 || s0) {
            uint32_t xs[W + 1];
#pragma unroll
            for (int w = 0; w <= W; ++w) xs[w] = b ? funnel(r[w + 1], r[w], 2 * b) : r[w];
            if (p0) {
                const uint32_t bp = to_vgpr_u((uint32_t)((sp.pre_pat >> (2 * b)) & 3ull) * 0x55555555u);
#pragma unroll
                for (int w = 0; w < W; ++w) { accP[w] |= xs[w] ^ bp; VF_KEEP(accP[w]); }
            }
            if (p1) {
                const uint32_t bp = to_vgpr_u((uint32_t)((sp.pre_pat >> (2 * b + 32)) & 3ull) * 0x55555555u);
#pragma unroll
                for (int w = 0; w < W; ++w) { accP[w] |= xs[w + 1] ^ bp; VF_KEEP(accP[w]); }
            }
            if (s0) {
                const uint32_t bp = to_vgpr_u((uint32_t)((sp.suf_pat >> (2 * b)) & 3ull) * 0x55555555u);
#pragma unroll
                for (int w = 0; w < W; ++w) { accS[w] |= xs[w] ^ bp; VF_KEEP(accS[w]); }
            }
            if (s1) {
                const uint32_t bp = to_vgpr_u((uint32_t)((sp.suf_pat >> (2 * b + 32)) & 3ull) * 0x55555555u);
#pragma unroll
                for (int w = 0; w < W; ++w) { accS[w] |= xs[w + 1] ^ bp; VF_KEEP(accS[w]); }
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
template <int W, bool P44>
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
        exact_find2<W, P44>(x, stg + lane_id(), n, sp, ps, ss);
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

// The same scan under INDEL distance (a mismatch = a deletion + an insertion = 2 units; round 4, DESIGN.md 4.3 "the indel scan").
// An acceptable alignment also has a bounded number of INDEL units (host: max_indel_units - the same knapsack with a
// substitution worth two), and where that bound is no larger than kerr the indel scan is the sharper filter at the same price:
// a copy with three substitutions has unit distance 3 but indel distance 6, and the scoring schemes that accept a five-base
// insertion (loss open + 4 ext) reject three substitutions (loss 3 (match - mismatch)).  At the cfg-4 shape (15 % substitutions)
// more than half of the tasks the unit-distance scan passed on were such copies: DP work that always ended in a rejection.
//
// Under indel distance neighbouring cells differ by exactly +-1 once the boundary is consistent, which the free-start row
// C[0][j] = 0 is not.  So the scan runs TWO lattices and takes the smaller value: sources restricted to even read positions
// (C[0][j] = j mod 2) and to odd ones (C[0][j] = (j + 1) mod 2; its phantom source before the read costs one unit more than
// the even lattice's source at 0 and never wins).  In each lattice, with a_i = the vertical step C[i][j-1] - C[i-1][j-1] and
// h_i = the horizontal step C[i][j] - C[i][j-1] (both +-1):  match: h_i = -a_i, v_i = -h_{i-1};  mismatch, a_i = -1: h_i = +1,
// v_i = -h_{i-1};  mismatch, a_i = +1: h_i = h_{i-1}, v_i = +1.  With A = {a_i = +1} that is a carry chain - generate ~A,
// propagate P = A & ~E - which ONE addition resolves: s = (~A | P) + ~A + h_0, carry into row i = s_i ^ P_i, and the new
// ~A is s & ~P.  State: nA = ~A per lattice (filler bits below the adapter: A = 1, E = 0, they hand the boundary step h_0 up to
// the first row).  C[m][j] = C[0][j] + m - 2 popcount(nA): one v_bcnt gives the detection.  5 instructions a lattice and
// column, 14 a column with the bookkeeping, against Myers' 15.  tests/tools/band_model.py restates it against the plain
// recurrence (indel_last_row); tests/test_band_model_cpu.py::test_indel_scan_bit_parallel_equals_the_recurrence.
template <uint32_t CIN> __device__ __forceinline__ void indel_column(uint32_t E, uint32_t &nA) {
    const uint32_t P = ~(nA | E);
    const uint32_t s = (P | nA) + nA + CIN;
    nA = s & ~P;
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
template <int W, bool INDEL>
__device__ __forceinline__ void myers_scan(const uint32_t *__restrict__ row, int n, const DpParams &dp, const uint2 *s_eq2,
                                           Clusters &cl, uint32_t (&xn)[10]) {
    constexpr int CW = 10;                             // words per chunk (xn holds chunk 0 on entry)
    uint32_t Pv = dp.pv_init, Mv = 0u;
    int sc = (int)dp.m - dp.kthr - 1;                  // unit edit distance minus (kthr + 1): negative <=> detected
    // indel scan: ~A of the even / odd lattice; a column is detected where popcount(nA) >= Q(C[0][j]), Q(c) = ceil((c + m - kthr) / 2);
    // w = popcount - Q per lattice and column, wmax0 / wmax1 its maximum over the cells with C[0][j] = 0 / 1 (-> dmin)
    uint32_t nAe = 0u, nAo = 0u;
    const int Q0 = ((int)dp.m - dp.kthr + 1) >> 1, Q1 = ((int)dp.m - dp.kthr + 2) >> 1;
    uint32_t negQ0, negQ1;                             // (VGPRs: a VALU instruction with an SGPR operand issues at half rate)
    asm("v_mov_b32 %0, %1" : "=v"(negQ0) : "s"(-Q0));
    asm("v_mov_b32 %0, %1" : "=v"(negQ1) : "s"(-Q1));
    int wmax0 = -Q0, wmax1 = -Q1, wown0 = -Q0, wown1 = -Q1;
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
                if constexpr (!INDEL) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        myers_column(e[q].x, Pv, Mv, sc); det = funnel(det, (uint32_t)sc, 31); smin = min(smin, sc);
                        myers_column(e[q].y, Pv, Mv, sc); det = funnel(det, (uint32_t)sc, 31); smin = min(smin, sc);
                        if (q & 1) smin_own = 16 * w + 2 * (q + 1) <= n ? smin : smin_own;
                    }
                } else {
                    asm volatile("" : "+v"(nAe), "+v"(nAo));
                    uint32_t nd = 0;                   // one NOT-detected bit per column (the sign of w in both lattices)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        // column 16 w + 2 q + 1 (odd): C[0][j] = 1 in the even lattice (it steps up: carry in), 0 in the odd one
                        indel_column<1u>(e[q].x, nAe); indel_column<0u>(e[q].x, nAo);
                        int we = (int)__builtin_popcount(nAe) + (int)negQ1, wo = (int)__builtin_popcount(nAo) + (int)negQ0;
                        nd = funnel(nd, (uint32_t)(we & wo), 31); wmax1 = max(wmax1, we); wmax0 = max(wmax0, wo);
                        // column 16 w + 2 q + 2 (even): the other way round
                        indel_column<0u>(e[q].y, nAe); indel_column<1u>(e[q].y, nAo);
                        we = (int)__builtin_popcount(nAe) + (int)negQ0; wo = (int)__builtin_popcount(nAo) + (int)negQ1;
                        nd = funnel(nd, (uint32_t)(we & wo), 31); wmax0 = max(wmax0, we); wmax1 = max(wmax1, wo);
                        if (q & 1) { const bool own = 16 * w + 2 * (q + 1) <= n; wown0 = own ? wmax0 : wown0; wown1 = own ? wmax1 : wown1; }
                    }
                    det = ~nd;
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
    if constexpr (!INDEL) cl.dmin = smin_own + dp.kthr + 1;
    else cl.dmin = min((int)dp.m - 2 * (wown0 + Q0), (int)dp.m + 1 - 2 * (wown1 + Q1));
}
// the scan the host chose for this adapter's scoring scheme (wave-uniform)
template <int W>
__device__ __forceinline__ void filter_scan(const uint32_t *__restrict__ row, int n, const DpParams &dp, const uint2 *s_eq2,
                                            Clusters &cl, uint32_t (&xn)[10]) {
    if (dp.indel) myers_scan<W, true>(row, n, dp, s_eq2, cl, xn);
    else myers_scan<W, false>(row, n, dp, s_eq2, cl, xn);
}

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
// Lay the adapter on a diagonal that ends in a column of the cluster - read bases [j - m, j), every j of the cluster (the
// kHamDiags centre-most of a wider one): a copy whose substitution sits in the adapter's last rows has a lopsided cluster (cutting
// those rows off is as cheap, in units, as the substitution) and its diagonal is not the centre one - and count the bases that
// differ; h = the smallest count.  That ungapped alignment scores match*m - h*(match - mismatch), so the task's optimum is
// S* >= T'' = max(T, that), and every OPTIMAL alignment has at most kq = max_unit_errors(T'') unit edits (host: fill_dp,
// dp.kerr_h[h]; kq >= h, so the certified column itself has D <= h <= kq).  The rest is the class-0 argument with kq in the
// place of kerr_lo: end columns of optimal alignments have D <= kq, D moves by at most one per column, so
// they lie kerr - kq columns (dp.dl_h[h]: the same difference in the scan's own units) inside the cluster [jmin, jmax] of
// columns with D <= kerr on every side the read does not cut off: [lo, hi]; alignments with at most kq gap or overhang bases ending there stay on the diagonals
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
    // every diagonal that ends in a column of the cluster, the kHamDiags centre-most if there are more: a window of the read that
    // slides by one base a step (three funnel shifts), not a gather per diagonal
    int j0 = cl.jmin1, cnt = cl.jmax1 - cl.jmin1 + 1;
    if (cnt > kHamDiags) { j0 += (cnt - kHamDiags) >> 1; cnt = kHamDiags; }
    const int dA = j0 - m;                             // first diagonal asked for (negative: the adapter would start left of the read)
    const int dS = cand ? max(dA, 0) : 0;              // first diagonal looked at
    const int idx = dS >> 4;
    const uint32_t bsh = 2u * (uint32_t)(dS & 15);
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (cand) {
        w0 = idx < W ? row[idx] : 0u;
        w1 = idx + 1 < W ? row[idx + 1] : 0u;
        w2 = idx + 2 < W ? row[idx + 2] : 0u;
        w3 = idx + 3 < W ? row[idx + 3] : 0u;
    }
    uint32_t a0 = funnel(w1, w0, bsh), a1 = funnel(w2, w1, bsh), a2 = funnel(w3, w2, bsh);      // read bases dS .. dS + 47
    const int dlast = dA + cnt - 1;
    int hmin = 64;
#pragma unroll
    for (int k = 0; k < kHamDiags; ++k) {
        const int d = dS + k;
        const bool fits = cand && d <= dlast && d + m <= n;       // asked for, and the whole adapter lies inside the read on it
        uint64_t diff = ((((uint64_t)a1 << 32) | a0) ^ dp.pat) & patmask;
        diff = (diff | (diff >> 1)) & 0x5555555555555555ull;
        const int h = (int)__popcll(diff);
        hmin = fits ? min(hmin, h) : hmin;
        a0 = funnel(a1, a0, 2); a1 = funnel(a2, a1, 2); a2 >>= 2;
    }
    int kq = -1;
#pragma unroll
    for (int h = 1; h < 8; ++h) kq = hmin == h ? dp.kerr_h[h] : kq;
    if (!cand || kq < 1 || kq >= dp.kerr) return;
    int dl = 0;                                        // how far inside the cluster the end columns of optimal alignments lie (host: fill_dp)
#pragma unroll
    for (int h = 1; h < 8; ++h) dl = hmin == h ? dp.dl_h[h] : dl;
    const int lo = cl.jmin1 > 1 ? cl.jmin1 + dl : 1;
    const int hi = cl.jmax1 < n ? cl.jmax1 - dl : n;
    if (hi < lo) return;
    const int width = hi - lo + 2 * kq + 1;
    if (width <= kBandN) { cls = 5; av = lo - m - kq; }         // (a single substitution: three diagonals)
    else if (width <= kBand) { cls = 0; av = lo - m - kq; }
    else if (width <= kBandW) { cls = 6; av = lo - m - kq; }
}

// What the Myers pass makes of one task's clusters: cls 0 banded (kBand), 1..3 windowed, 4 whole read, 5 narrow band (kBandN), 6 wide band (kBandW);
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
        const int dl = dp.dl_lo;
        const int lo = cl.jmin1 > 1 ? cl.jmin1 + dl : 1;
        const int hi = cl.jmax1 < n ? cl.jmax1 - dl : n;
        if (hi >= lo && hi - lo + 2 * dp.kerr_lo + 1 <= kBand) { cls = hi - lo + 2 * dp.kerr_lo + 1 <= kBandN ? 5 : 0; av = lo - (int)dp.m - dp.kerr_lo; }
    }
    if (VF_HAMMING_CERT) hamming_band_class<W>(row, n, cl, dp, any && cls != 0 && cls != 5 && cl.ncl == 1 && dp.kerr_h[0] != 0, cls, av);
    if (cls != 0 && cls != 5 && cls != 6 && wide_band_class(cl, dp, av)) cls = 6;
}

// the Myers pass alone, over every task of the adapter's list (scoring schemes that give the block filter no bound, or
// VFIND_BLOCK_FILTER=0)
template <int W>
__global__ __launch_bounds__(kBlock) void k_filter(DpParams dp, short2 *__restrict__ spans, BatchLists bl,
                                                   uint64_t *__restrict__ counters) {
    const TaskSrc src{bl.rows, bl.lens, W};
    __shared__ uint32_t s_c[7];
    __shared__ uint2 s_eq2[16];
    if (threadIdx.x < 16) s_eq2[threadIdx.x] = make_uint2(dp.eq_top[threadIdx.x & 3], dp.eq_top[threadIdx.x >> 2]);
    const uint32_t *list = dp.is_prefix ? bl.listP : bl.listS;
    const uint32_t *cnt = dp.is_prefix ? bl.cntP : bl.cntS;
    uint32_t stat = 0;
    for (uint32_t b = blockIdx.x; b < bl.grid; b += gridDim.x) {
        const uint32_t ntask = cnt[b];
        if (threadIdx.x < 7) s_c[threadIdx.x] = 0u;
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
            filter_scan<W>(task_row(src, r), n, dp, s_eq2, cl, x0);
            const bool any = valid && cl.ncl > 0;
            if (valid && !any) (void)dp_commit(true, task_head(src, r), false, 0, dp, spans);      // (the header again: nothing of it is kept across the scan)
            int cls, av, a1, a2;
            myers_classify<W>(task_row(src, r), n, any, cl, dp, cls, av, a1, a2);
            {
                const bool toB = any && cls == 6;
                const uint32_t ib = wave_append(&s_c[5], toB);
                if (toB) { bl.listB[seg0 + ib] = r; bl.winB[seg0 + ib] = (uint32_t)(av + kBandBias); }
                const bool toN = any && cls == 5;
                const uint32_t in = wave_append(&s_c[6], toN);
                if (toN) { bl.listN[seg0 + in] = r; bl.winN[seg0 + in] = (uint32_t)(av + kBandBias); }
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
        if (threadIdx.x == 5) { bl.cntN[b] = s_c[6]; if (s_c[6]) atomicAdd(&bl.tot[(dp.is_prefix ? 0u : 8u) + 5u], s_c[6]); }
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
                                                  uint32_t *s_my, const uint2 *s_eq2, uint32_t *s_narrow) {
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
        filter_scan<W>(task_row(src, r), n, dp, s_eq2, cl, x0);
        const bool any = valid && cl.ncl > 0;
        if (valid && !any) (void)dp_commit(true, task_head(src, r), false, 0, dp, spans);      // (the header again: nothing of it is kept across the scan)
        int cls, av, a1, a2;
        myers_classify<W>(task_row(src, r), n, any, cl, dp, cls, av, a1, a2);
        {
            const bool toB = any && cls == 6;
            const uint32_t ib = wave_append(&s_my[5], toB);
            if (toB) { cs.listB[seg0 + ib] = r; cs.winB[seg0 + ib] = (uint32_t)(av + kBandBias); }
            const bool toN = any && cls == 5;                   // behind the narrow tasks the block filter put into the segment
            const uint32_t in = wave_append(s_narrow, toN);
            if (toN) { cs.listN[seg0 + in] = r; cs.winN[seg0 + in] = (uint32_t)(av + kBandBias); }
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
#ifndef VF_BMF_MINBLOCKS
#define VF_BMF_ATTR __launch_bounds__(kBlock)
#else
#define VF_BMF_ATTR __launch_bounds__(kBlock, (W == 10 ? VF_BMF_MINBLOCKS : 1))     // A/B: register cap for the 150 bp instances (5 -> 96 registers: five waves a SIMD)
#endif
template <int W, int NB>
__global__ VF_BMF_ATTR void k_bmfilter(DpParams dp,
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
        if (nR) front_myers_stage<W>(src, dp, spans, cs, nR, seg0, s_my, s_eq2, &s_bf[0]);
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

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <int W> static int scan_occupancy() {
    // the list segments k_scan writes are read by one workgroup each of the filter kernels: the same count has to suit them
    int o = std::min(occupancy_of(k_scan<W, false>, 0), occupancy_of(k_scan<W, true>, 0));
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

hipError_t launch_scan(hipStream_t s, int W, const uint32_t *rows, const uint16_t *lens, uint64_t n, const ScanParams &sp,
                       short2 *spans, BatchLists bl) {
    if (n == 0) return hipSuccess;
    uint32_t g, seg;
    scan_geometry(n, W, &g, &seg);
    if (bl.grid != g || bl.seg != seg) return hipErrorInvalidValue;
    const bool p44 = sp.pre_len >= 20 && sp.suf_len >= 20;        // scan_plan(m) = {4, 4} for every m >= 20
    switch (W) {
    case 10: if (p44) hipLaunchKernelGGL((k_scan<10, true>), dim3(g), dim3(kBlock), 0, s, rows, lens, n, sp, spans, bl);
             else hipLaunchKernelGGL((k_scan<10, false>), dim3(g), dim3(kBlock), 0, s, rows, lens, n, sp, spans, bl);
             break;
    case 20: if (p44) hipLaunchKernelGGL((k_scan<20, true>), dim3(g), dim3(kBlock), 0, s, rows, lens, n, sp, spans, bl);
             else hipLaunchKernelGGL((k_scan<20, false>), dim3(g), dim3(kBlock), 0, s, rows, lens, n, sp, spans, bl);
             break;
    case 40: if (p44) hipLaunchKernelGGL((k_scan<40, true>), dim3(g), dim3(kBlock), 0, s, rows, lens, n, sp, spans, bl);
             else hipLaunchKernelGGL((k_scan<40, false>), dim3(g), dim3(kBlock), 0, s, rows, lens, n, sp, spans, bl);
             break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
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



}  // namespace vf
