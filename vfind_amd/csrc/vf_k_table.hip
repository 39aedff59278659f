// vf_k_table.hip -- k_finish (extract, translate, count), the count log, k_generic, table export / merge / growth / clear
// (kernels of the variant-extraction hot path, gfx950; the map of the translation units is in vf_kernels_common.h)
#include "vf_kernels_common.h"
#include "vf_table_dev.h"

namespace vf {

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
