// Micro-benchmark 3: issue cost of the 3-operand and carry instructions the filter / scan kernels use.
// hipcc --offload-arch=gfx950 -O3 valu_rate3.hip -o valu_rate3 && ./valu_rate3
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t s0, uint32_t s1) {
    uint32_t a = threadIdx.x, b = threadIdx.x * 3 + 1, c = threadIdx.x ^ 5, d = s0 + threadIdx.x, e = 7, f = 9;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %1\n v_add_u32 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 1) { REP16(asm volatile("v_bitop3_b32 %0, %0, %1, %3 bitop3:0xde\n v_bitop3_b32 %2, %2, %3, %1 bitop3:0xde\n v_bitop3_b32 %4, %4, %1, %3 bitop3:0xde\n v_bitop3_b32 %5, %5, %3, %1 bitop3:0xde" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 2) { REP16(asm volatile("v_bitop3_b32 %0, %0, %1, %6 bitop3:0xde\n v_bitop3_b32 %2, %2, %3, %6 bitop3:0xde\n v_bitop3_b32 %4, %4, %1, %6 bitop3:0xde\n v_bitop3_b32 %5, %5, %3, %6 bitop3:0xde" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(s1));) }
        if (OP == 3) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_add_co_u32 %4, vcc, %4, %4\n v_subbrev_co_u32 %2, vcc, 0, %2, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");) }
        if (OP == 4) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_add_co_u32 %2, vcc, %2, %3\n v_add_co_u32 %4, vcc, %4, %1\n v_add_co_u32 %5, vcc, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");) }
        if (OP == 5) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %1, vcc\n v_cndmask_b32 %5, %5, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 6) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 1, %1\n v_lshl_or_b32 %2, %2, 1, %3\n v_lshl_or_b32 %4, %4, 1, %1\n v_lshl_or_b32 %5, %5, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 7) { REP16(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %2, 5\n v_readlane_b32 s22, %4, 7\n v_readlane_b32 s23, %5, 9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "s20", "s21", "s22", "s23");) }
        if (OP == 8) { REP16(asm volatile("v_writelane_b32 %0, %6, 3\n v_writelane_b32 %2, %6, 5\n v_writelane_b32 %4, %6, 7\n v_writelane_b32 %5, %6, 9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(s1));) }
        if (OP == 9) { REP16(asm volatile("v_xor_b32 %0, %0, %1\n v_or_b32 %2, %2, %0\n v_xor_b32 %4, %4, %3\n v_or_b32 %5, %5, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 10) { REP16(asm volatile("v_max_i32 %0, %0, %1\n v_max_i32 %2, %2, %3\n v_max_i32 %4, %4, %1\n v_max_i32 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 11) { REP16(asm volatile("v_alignbit_b32 %0, %1, %0, 2\n v_alignbit_b32 %2, %3, %2, 2\n v_alignbit_b32 %4, %1, %4, 2\n v_alignbit_b32 %5, %3, %5, 2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 12) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %1\n v_mov_b32 %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 13) { REP16(asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %2, %2, %3\n v_pk_max_i16 %4, %4, %1\n v_pk_max_i16 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 14) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_bitop3_b32 %2, %2, %3, %1 bitop3:0xde\n v_add_u32 %4, %4, %1\n v_bitop3_b32 %5, %5, %3, %1 bitop3:0xde" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 15) { REP16(asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %3, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %3, %5 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 16) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cndmask_b32 %4, %4, %1, s[20:21]\n v_cndmask_b32 %5, %5, %3, s[20:21]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "s20", "s21");) }
        if (OP == 17) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u32 vcc, %4, %1\n v_cndmask_b32 %5, %5, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");) }
        if (OP == 18) { REP16(asm volatile("v_xor_b32 %0, %6, %0\n v_or_b32 %2, %6, %2\n v_and_b32 %4, %6, %4\n v_add_u32 %5, %6, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(s1));) }
        if (OP == 19) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %4, %1\n v_cmp_lt_u32 vcc, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");) }
        if (OP == 20) { REP16(asm volatile("v_bfe_i32 %0, %1, 3, 1\n v_bfe_i32 %2, %3, 5, 1\n v_bfe_i32 %4, %1, 7, 1\n v_bfe_i32 %5, %3, 9, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 21) { REP16(asm volatile("v_lshrrev_b32 %0, %6, %0\n v_lshlrev_b32 %2, %6, %2\n v_ashrrev_i32 %4, %6, %4\n v_lshrrev_b32 %5, %6, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(s1));) }
        if (OP == 22) { REP16(asm volatile("v_lshrrev_b32 %0, %1, %0\n v_lshlrev_b32 %2, %3, %2\n v_ashrrev_i32 %4, 31, %4\n v_lshrrev_b32 %5, 3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 23) { REP16(asm volatile("v_sub_u32 %0, %0, %1\n v_subrev_u32 %2, %2, %3\n v_not_b32 %4, %4\n v_xnor_b32 %5, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 24) { REP16(asm volatile("v_ffbh_u32 %0, %1\n v_ffbl_b32 %2, %3\n v_bcnt_u32_b32 %4, %1, %4\n v_mbcnt_lo_u32_b32 %5, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 25) { REP16(asm volatile("v_add3_u32 %0, %0, %1, %3\n v_add3_u32 %2, %2, %3, %1\n v_max3_i32 %4, %4, %1, %3\n v_max3_i32 %5, %5, %3, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 26) { REP16(asm volatile("v_and_b32 %0, 0x78, %0\n v_and_b32 %2, 0xffff, %2\n v_or_b32 %4, 0x12345678, %4\n v_add_u32 %5, 0x55555555, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f;
}

template <int OP> void run(const char *name, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 1000;
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
    printf("%-34s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD\n", name, blocks_per_cu, ms, 1e9 / per_simd_per_s);
    hipFree(out);
}

int main() {
    for (int occ : {8}) {
        run<16>("v_cndmask_b32 (sgpr pair mask)", occ); run<17>("v_cmp + v_cndmask pairs", occ); run<18>("xor/or/and/add with sgpr src0", occ); run<19>("v_cmp_lt_u32", occ);
        run<20>("v_bfe_i32", occ); run<21>("shifts by sgpr", occ); run<22>("shifts by vgpr / const", occ); run<23>("sub/subrev/not/xnor", occ); run<24>("ffbh/ffbl/bcnt/mbcnt", occ);
        run<25>("add3 / max3", occ); run<26>("vop2 with literal", occ);
        printf("\n");
    }
    for (int occ : {8}) {
        run<0>("v_add_u32", occ); run<1>("v_bitop3_b32 (vvv)", occ); run<2>("v_bitop3_b32 (vvs)", occ); run<3>("add_co/addc/add_co/subbrev", occ);
        run<4>("v_add_co_u32", occ); run<5>("v_cndmask_b32", occ); run<6>("v_lshl_or_b32", occ); run<7>("v_readlane_b32", occ); run<8>("v_writelane_b32", occ);
        run<9>("xor,or dependent pairs", occ); run<10>("v_max_i32", occ); run<11>("v_alignbit_b32", occ); run<12>("v_mov_b32", occ); run<13>("v_pk_add_u16/v_pk_max_i16", occ);
        run<14>("add + bitop3 alternating", occ); run<15>("v_add_u32_dpp", occ);
        printf("\n");
    }
    return 0;
}
