// vf_pack.cpp -- ASCII -> 2-bit packed rows on the host (host-only translation unit: x86 intrinsics)
#include "../../include/vfind_hip.h"

#include <immintrin.h>

#include <cstring>
#include <mutex>

// ------------------------------------------------------------------------------------------------
// packing
// ------------------------------------------------------------------------------------------------
static uint8_t g_code[256];
static std::once_flag g_code_once;
static void init_code() {
    memset(g_code, 0xFF, sizeof(g_code));
    g_code[(int)'A'] = 0; g_code[(int)'C'] = 1; g_code[(int)'G'] = 2; g_code[(int)'T'] = 3;
}

extern "C" uint32_t vf_row_dwords_for(uint32_t max_len) {
    if (max_len <= 160) return 10;
    if (max_len <= 320) return 20;
    if (max_len <= 640) return 40;
    return 0;
}

// 2-bit pack; returns 1 when a byte outside {A,C,G,T} is met (row content then undefined)
static int pack_read_scalar(const uint8_t *seq, uint32_t n, uint32_t *row, uint32_t row_dwords) {
    uint32_t bad = 0;
    uint32_t full = n / 16;
    for (uint32_t w = 0; w < full; ++w) {
        const uint8_t *p = seq + 16 * w;
        uint32_t v = 0;
        for (int k = 0; k < 16; ++k) {
            const uint32_t c = g_code[p[k]];
            bad |= c;
            v |= (c & 3u) << (2 * k);
        }
        row[w] = v;
    }
    uint32_t w = full;
    if (n % 16) {
        const uint8_t *p = seq + 16 * w;
        uint32_t v = 0;
        for (uint32_t k = 0; k < n % 16; ++k) {
            const uint32_t c = g_code[p[k]];
            bad |= c;
            v |= (c & 3u) << (2 * k);
        }
        row[w++] = v;
    }
    for (; w < row_dwords; ++w) row[w] = 0;
    return (bad & 0x80u) ? 1 : 0;
}

// 16 bases a step: membership in {A,C,G,T} by two nibble look-ups (A,C,G = 0x4{1,3,7}, T = 0x54), the code
// as x ^ (x >> 1) with x = bits 2:1 of the letter (A 00, C 01, G 11, T 10 -> 0,1,2,3), sixteen 2-bit fields
// gathered by two PEXTs.  The per-byte table loop above cost the frame + pack workers ~200 ns per read.
__attribute__((target("ssse3,bmi2"))) static inline uint32_t pack16(__m128i v, __m128i &bad) {
    const __m128i lo_tab = _mm_setr_epi8(0, 1, 0, 1, 2, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0);       // low nibble -> classes it is legal in
    const __m128i hi_tab = _mm_setr_epi8(0, 0, 0, 0, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);       // high nibble -> its class
    const __m128i nib = _mm_set1_epi8(0x0F), three = _mm_set1_epi8(3);
    const __m128i ok = _mm_and_si128(_mm_shuffle_epi8(lo_tab, _mm_and_si128(v, nib)), _mm_shuffle_epi8(hi_tab, _mm_and_si128(_mm_srli_epi16(v, 4), nib)));
    bad = _mm_or_si128(bad, _mm_cmpeq_epi8(ok, _mm_setzero_si128()));
    const __m128i x = _mm_and_si128(_mm_srli_epi16(v, 1), three);
    const __m128i code = _mm_xor_si128(x, _mm_and_si128(_mm_srli_epi16(x, 1), three));
    const uint64_t lo = (uint64_t)_mm_cvtsi128_si64(code), hi = (uint64_t)_mm_cvtsi128_si64(_mm_srli_si128(code, 8));
    return (uint32_t)_pext_u64(lo, 0x0303030303030303ull) | ((uint32_t)_pext_u64(hi, 0x0303030303030303ull) << 16);
}
__attribute__((target("ssse3,bmi2"))) static int pack_read_simd(const uint8_t *seq, uint32_t n, uint32_t *row, uint32_t row_dwords) {
    __m128i bad = _mm_setzero_si128();
    const uint32_t full = n / 16;
    uint32_t w = 0;
    for (; w < full; ++w) row[w] = pack16(_mm_loadu_si128((const __m128i *)(seq + 16 * w)), bad);
    if (n % 16) {
        uint8_t tail[16];
        memset(tail, 'A', sizeof(tail));                   // padding packs to 0, as the row's zero fill
        memcpy(tail, seq + 16 * w, n % 16);
        row[w++] = pack16(_mm_loadu_si128((const __m128i *)tail), bad);
    }
    for (; w < row_dwords; ++w) row[w] = 0;
    return _mm_movemask_epi8(bad) ? 1 : 0;
}

extern "C" int vf_pack_read(const uint8_t *seq, uint32_t n, uint32_t *row, uint32_t row_dwords) {
    std::call_once(g_code_once, init_code);
    if (n > row_dwords * 16u) return 1;
    static const bool simd = __builtin_cpu_supports("ssse3") && __builtin_cpu_supports("bmi2");
    return simd ? pack_read_simd(seq, n, row, row_dwords) : pack_read_scalar(seq, n, row, row_dwords);
}

