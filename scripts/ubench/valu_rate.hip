// Micro-benchmark: sustained issue rate of the integer VALU instructions the DP kernels use.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t s0, uint32_t s1) {
    uint32_t a = threadIdx.x, b = threadIdx.x * 3 + 1, c = threadIdx.x ^ 5, d = s0 + threadIdx.x, e = 7, f = 9, g = 11, h = 13;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %1\n v_add_u32 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 1) { REP16(asm volatile("v_max_i32 %0, %0, %1\n v_max_i32 %2, %2, %3\n v_max_i32 %4, %4, %1\n v_max_i32 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 2) { REP16(asm volatile("v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %3, %3, %1, %2\n v_max3_i32 %4, %4, %1, %2\n v_max3_i32 %5, %5, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 3) { REP16(asm volatile("v_mad_u32_u24 %0, %1, %6, %0\n v_mad_u32_u24 %2, %3, %6, %2\n v_mad_u32_u24 %4, %1, %6, %4\n v_mad_u32_u24 %5, %3, %6, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(s1));) }
        if (OP == 4) { REP16(asm volatile("v_and_or_b32 %0, %0, %6, %1\n v_and_or_b32 %2, %2, %6, %3\n v_and_or_b32 %4, %4, %6, %1\n v_and_or_b32 %5, %5, %6, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(s1));) }
        if (OP == 5) { REP16(asm volatile("v_bfe_u32 %0, %1, 3, 1\n v_bfe_u32 %2, %3, 5, 1\n v_bfe_u32 %4, %1, 7, 1\n v_bfe_u32 %5, %3, 9, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 6) { REP16(asm volatile("v_alignbit_b32 %0, %1, %0, 2\n v_alignbit_b32 %2, %3, %2, 2\n v_alignbit_b32 %4, %1, %4, 2\n v_alignbit_b32 %5, %3, %5, 2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 7) { REP16(asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %2, 0x54321, %2\n v_and_b32 %4, 0xffffe3ff, %4\n v_or_b32 %5, 0x400, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 8) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }   // dependent chain
        if (OP == 9) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %4, %4, %1\n v_mul_lo_u32 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 10) { REP16(asm volatile("v_xor_b32 %0, %0, %1\n v_or_b32 %2, %2, %3\n v_and_b32 %4, %4, %1\n v_lshlrev_b32 %5, 1, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;
}

template <int OP> void run(const char *name, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 2000;
    uint32_t *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 1u, 3u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u, 3u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * 4 /*waves*/ * iters * 64.0;     // wave-instructions
    const double per_simd_per_s = instr / (ms * 1e-3) / 1024.0;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f G wave-instr/s/SIMD  (%.2f cycles/instr at 2.4 GHz)  lane-ops %.1f T/s\n", name, blocks_per_cu,
           ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, instr * 64 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int occ : {1, 2, 4, 8}) {
        run<0>("v_add_u32", occ); run<1>("v_max_i32", occ); run<2>("v_max3_i32", occ); run<3>("v_mad_u32_u24 (sgpr)", occ);
        run<4>("v_and_or_b32 (sgpr)", occ); run<5>("v_bfe_u32", occ); run<6>("v_alignbit_b32", occ); run<7>("vop2 with literal", occ);
        run<8>("v_add_u32 dependent", occ); run<9>("v_mul_lo_u32", occ); run<10>("xor/or/and/lshl", occ);
        printf("\n");
    }
    return 0;
}
