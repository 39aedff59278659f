// How many memory-side requests does a stream of 40-byte rows cost, by the shape of the loads?
// (the question behind profiles/r04_requests.json: k_scan moves 585 MB through the L2's fabric interface for 420 MB of rows)
//
// Every variant reads the same n rows of 10 dwords (rows 8-byte aligned, as the packed batches are) and xors them into one word
// per lane, grid-stride over 1280 workgroups of 256 lanes - k_scan's geometry.
//   mode 0: one row per lane, five 8-byte loads                      (load_row as k_scan / k_finish have it)
//   mode 1: one row per lane, 16 + 16 + 8 byte loads
//   mode 2: the wave's 2 560 contiguous bytes by coalesced 16-byte loads (lane l takes bytes 16 l .. of each kilobyte), 2.5 instructions
//   mode 3: mode 0 with the NEXT trip's row fetched before the current one is used (k_scan's prefetch)
// run under rocprofv3 --pmc TCC_EA0_RDREQ_sum (and TCC_REQ_sum): per-launch requests per mode; plain run: GB/s per mode.
// build: hipcc --offload-arch=gfx950 -O2 -o row_stream row_stream.hip      run: ./row_stream [rows=10000000] [reps=5]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

constexpr int kBlock = 256, W = 10;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(kBlock) void k_stream(const uint32_t *__restrict__ rows, uint64_t n, uint32_t *__restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint32_t acc = 0;
    if (MODE == 0 || MODE == 3) {
        uint32_t nx[W];
        auto load = [&](uint64_t r, uint32_t (&x)[W]) {
            const u32x2 *p = reinterpret_cast<const u32x2 *>(rows + r * W);
#pragma unroll
            for (int i = 0; i < W / 2; ++i) { const u32x2 v = p[i]; x[2 * i] = v.x; x[2 * i + 1] = v.y; }
        };
        uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
        if (MODE == 3 && r < n) load(r, nx);
        for (; r < n; r += stride) {
            uint32_t x[W];
            if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < W; ++i) x[i] = nx[i];
                if (r + stride < n) load(r + stride, nx);
            } else load(r, x);
#pragma unroll
            for (int i = 0; i < W; ++i) acc ^= x[i] * (uint32_t)(i + 1);
        }
    } else if (MODE == 1) {
        for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride) {
            const uint32_t *p = rows + r * W;
            u32x4 a, b; u32x2 c;
            // (rows are 8-byte aligned only: unaligned 16-byte global loads are legal on gfx950)
            asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\tglobal_load_dwordx2 %2, %3, off offset:32\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p) : "memory");
            acc ^= a.x ^ a.y * 2 ^ a.z * 3 ^ a.w * 4 ^ b.x * 5 ^ b.y * 6 ^ b.z * 7 ^ b.w * 8 ^ c.x * 9 ^ c.y * 10;
        }
    } else {
        // the wave's 64 rows = 2 560 bytes = 160 x 16 bytes: lanes take consecutive 16-byte pieces
        const uint32_t lane = threadIdx.x & 63u;
        for (uint64_t r0 = ((uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u)); r0 < n; r0 += stride) {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(rows + r0 * W);
            const uint64_t pieces = (n - r0 >= 64 ? 64 : n - r0) * W / 4;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t i = lane + 64u * (uint32_t)k;
                if (i < pieces) { const u32x4 v = p[i]; acc ^= v.x ^ v.y * 2 ^ v.z * 3 ^ v.w * 4; }
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : 10000000ull;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    uint32_t *rows, *out;
    hipMalloc(&rows, n * W * 4 + 64);
    hipMalloc(&out, 64);
    hipMemset(rows, 0x5A, n * W * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int r = 0; r < reps; ++r) {
            hipEventRecord(a, 0);
            switch (mode) {
            case 0: hipLaunchKernelGGL(k_stream<0>, dim3(1280), dim3(kBlock), 0, 0, rows, n, out); break;
            case 1: hipLaunchKernelGGL(k_stream<1>, dim3(1280), dim3(kBlock), 0, 0, rows, n, out); break;
            case 2: hipLaunchKernelGGL(k_stream<2>, dim3(1280), dim3(kBlock), 0, 0, rows, n, out); break;
            default: hipLaunchKernelGGL(k_stream<3>, dim3(1280), dim3(kBlock), 0, 0, rows, n, out); break;
            }
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("mode %d: %.3f ms, %.0f GB/s\n", mode, best, n * W * 4 / (best * 1e-3) / 1e9);
    }
    return 0;
}
