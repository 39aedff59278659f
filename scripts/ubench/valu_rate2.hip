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

        if (OP == 11) { REP16(asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %2, %2, %3\n v_max_f32 %4, %4, %1\n v_max_f32 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 12) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %1, vcc\n v_cndmask_b32 %5, %5, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");) }
        if (OP == 13) { REP16(asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cmp_eq_u32 vcc, %2, %3\n v_cmp_gt_i32 vcc, %4, %1\n v_cmp_gt_i32 vcc, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");) }
        if (OP == 14) { REP16(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %3, %3, %1, %2\n v_add3_u32 %4, %4, %1, %2\n v_add3_u32 %5, %5, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 15) { REP16(asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %2, %2, %3\n v_pk_max_i16 %4, %4, %1\n v_pk_max_i16 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 16) { REP16(asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %4, %4, %1\n v_pk_add_u16 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 17) { REP16(asm volatile("v_add_u32_e64 %0, %0, %1\n v_add_u32_e64 %2, %2, %3\n v_add_u32_e64 %4, %4, %1\n v_add_u32_e64 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 18) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %4, %4, %1\n v_mul_u32_u24 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 19) { REP16(asm volatile("v_sub_u32 %0, %0, %1\n v_lshlrev_b32 %2, 3, %2\n v_ashrrev_i32 %4, 31, %4\n v_lshrrev_b32 %5, 2, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 20) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %1\n v_mov_b32 %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 21) { REP16(asm volatile("v_min_u32 %0, %0, %1\n v_min_i32 %2, %2, %3\n v_max_u32 %4, %4, %1\n v_min_u32 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 22) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 2, %1\n v_lshl_or_b32 %2, %2, 3, %3\n v_bfi_b32 %4, %1, %4, %3\n v_perm_b32 %5, %5, %3, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 23) { REP16(asm volatile("v_max_f16 %0, %0, %1\n v_pk_max_f16 %2, %2, %3\n v_max_f16 %4, %4, %1\n v_pk_max_f16 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 24) { REP16(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %3, %3, %1, %2\n v_max3_f32 %4, %4, %1, %2\n v_max3_f32 %5, %5, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 25) { REP16(asm volatile("v_sub_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_fma_f32 %4, %4, %1, %3\n v_mul_f32 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 26) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %6\n v_cndmask_b32_e64 %2, %2, %3, %6\n v_cndmask_b32_e64 %4, %4, %1, %6\n v_cndmask_b32_e64 %5, %5, %3, %6" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"((unsigned long long)s1 * 0x100000001ull));) }
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
    for (int occ : {2, 8}) {
        run<0>("v_add_u32", occ); run<1>("v_max_i32", occ); run<11>("v_max_f32", occ); run<12>("v_cndmask_b32 vcc", occ); run<26>("v_cndmask_b32_e64 sgpr", occ);
        run<13>("v_cmp", occ); run<14>("v_add3_u32", occ); run<15>("v_pk_max_i16", occ); run<16>("v_pk_add_u16", occ); run<17>("v_add_u32_e64", occ);
        run<18>("v_mul_u32_u24", occ); run<19>("sub/lshl/ashr/lshr", occ); run<20>("v_mov_b32", occ); run<21>("v_min/max u32/i32", occ);
        run<22>("lshl_add/lshl_or/bfi/perm", occ); run<23>("max_f16/pk_max_f16", occ); run<24>("v_max3_f32", occ); run<25>("f32 add/sub/fma/mul", occ);
        printf("\n");
    }
    return 0;
}
