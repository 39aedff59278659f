// vf_k_text.hip -- FASTQ text on the device: newline index, four-line framing, 2-bit packing (vf_run_fastq's device framing)
// (kernels of the variant-extraction hot path, gfx950; the map of the translation units is in vf_kernels_common.h)
#include "vf_kernels_common.h"

namespace vf {

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
// launchers
// ------------------------------------------------------------------------------------------------
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


}  // namespace vf
