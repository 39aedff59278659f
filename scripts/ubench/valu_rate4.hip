// Micro-benchmark 4: does compiler-generated straight-line code reach the issue rate the asm loops reach?
// hipcc --offload-arch=gfx950 -O3 valu_rate4.hip -o valu_rate4
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
__device__ __forceinline__ uint32_t to_vgpr_u(uint32_t s) { uint32_t v; asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v; }

// OP 0: scan stage 1 as in vf_k_front.hip (4 shifts, 8 bases, 2 adapters), data in registers
// OP 1: the same without the funnel shifts (bitop3 only)
// OP 2: only the funnel shifts
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint64_t pp, uint64_t sp) {
    constexpr int W = 10;
    uint32_t r[W + 2];
#pragma unroll
    for (int i = 0; i < W + 2; ++i) r[i] = threadIdx.x * 2654435761u + i * 40503u;
    uint32_t accP[W], accS[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { accP[w] = 0; accS[w] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            uint32_t xs[W + 1];
#pragma unroll
            for (int w = 0; w <= W; ++w) xs[w] = (OP == 1) ? r[w] : (b ? funnel(r[w + 1], r[w], 2 * b) : r[w]);
            if (OP == 2) {
#pragma unroll
                for (int w = 0; w < W; ++w) accP[w] += xs[w];
                accS[0] += xs[W];
            } else {
                const uint32_t bp0 = to_vgpr_u((uint32_t)((pp >> (2 * b)) & 3ull) * 0x55555555u);
                const uint32_t bp1 = to_vgpr_u((uint32_t)((pp >> (2 * b + 32)) & 3ull) * 0x55555555u);
                const uint32_t bs0 = to_vgpr_u((uint32_t)((sp >> (2 * b)) & 3ull) * 0x55555555u);
                const uint32_t bs1 = to_vgpr_u((uint32_t)((sp >> (2 * b + 32)) & 3ull) * 0x55555555u);
#pragma unroll
                for (int w = 0; w < W; ++w) accP[w] |= xs[w] ^ bp0;
#pragma unroll
                for (int w = 0; w < W; ++w) accP[w] |= xs[w + 1] ^ bp1;
#pragma unroll
                for (int w = 0; w < W; ++w) accS[w] |= xs[w] ^ bs0;
#pragma unroll
                for (int w = 0; w < W; ++w) accS[w] |= xs[w + 1] ^ bs1;
            }
        }
        // keep the loop from collapsing: rotate the row by one word and fold the accumulators in
#pragma unroll
        for (int w = 0; w < W; ++w) { r[w] = r[w + 1] + (accP[w] & 1u); accP[w] >>= 1; accS[w] >>= 1; }
        r[W] = r[0] ^ accS[3];
    }
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) s += accP[w] ^ accS[w] ^ r[w];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> void run(const char *name, int blocks_per_cu, double instr_per_iter) {
    const int blocks = 256 * blocks_per_cu, iters = 4000;
    uint32_t *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 0x123456789abcdefull, 0xfedcba987654321ull);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 0x123456789abcdefull, 0xfedcba987654321ull);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * 4 * iters * instr_per_iter;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD (assuming %.0f VALU instr / trip)\n", name, blocks_per_cu, ms,
           ms * 1e6 / (instr / 1024.0), instr_per_iter);
    hipFree(out);
}

int main() {
    for (int occ : {2, 8}) {
        run<0>("scan stage 1 (4 shifts)", occ, 3 * 11 + 160 + 31);
        run<1>("bitop3 only", occ, 160 + 31);
        run<2>("alignbit + add", occ, 3 * 11 + 44 + 31);
    }
    return 0;
}
