// Does a reader on another XCD that sees a flag also see the payload its writer stored BEFORE the flag?
// (the question behind DESIGN.md section 9, round 3: a LONG slot line's offset word against its READY bit)
//
// One 64-byte line per (writer lane, reader lane) pair and step, the pair always on different XCDs (workgroup b runs on XCD
// b mod 8: writers are the even workgroups, readers the odd ones).  A line starts a round holding the previous round's
// flag word and payload (both non-zero, both different from this round's).  The writer claims the flag word by CAS (no
// READY), stores the payload, waits for the store (s_waitcnt vmcnt(0)), stores flag | READY - the count table's
// protocol for a LONG key.  The reader spins on ONE 16-byte load {flag, payload} until it sees this round's flag with READY
// and looks at the payload that came with it.
//   mode 0: as above (two 8-byte agent-scope stores with a wait between them)
//   mode 1: the same with an agent-scope release fence before the flag store
//   mode 2: payload and flag | READY in one 16-byte store
//   mode 3: mode 0, payload in the line's second 16 bytes (two loads on the reader's side: flag first, payload second)
// Output per mode: transitions seen, torn ones (READY with the OLD payload), readers that gave up.
// MI355X, 2^22 lines x 40 rounds: 167.6-167.8 M transitions watched per mode, 0 torn in every mode (12 / 45 / 11 / 11 ms):
// the hardware keeps store -> wait -> flag in order across XCDs.  (What had looked like a torn line in the count table was
// a w0 returned by a lost CAS judged against the line image loaded before the CAS: DESIGN.md section 9.)
// build: hipcc --offload-arch=gfx950 -O2 -o visibility visibility.hip     run: ./visibility [lines_log2=22] [rounds=40]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

constexpr int kBlock = 256;
constexpr unsigned long long kReady = 1ull << 63;

struct Line { unsigned long long w[8]; };

__device__ __forceinline__ unsigned long long flag_of(uint64_t round, uint64_t i) { return ((round & 0x7FFFFFull) << 40) | (i & 0xFFFFFFFFFFull) | (1ull << 62); }
__device__ __forceinline__ unsigned long long payload_of(uint64_t round, uint64_t i) { return 0x4C4C4C4C00000000ull ^ (round << 33) ^ (i * 0x9E3779B97F4A7C15ull >> 8) | 1ull; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 load16(const void *p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store16(void *p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(kBlock) void k_init(Line *lines, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        Line l;
        for (int k = 0; k < 8; ++k) l.w[k] = 0x8080808080808080ull;
        l.w[0] = flag_of(0, i) | kReady; l.w[1] = payload_of(0, i); l.w[2] = payload_of(0, i);
        lines[i] = l;
    }
}

__global__ __launch_bounds__(kBlock) void k_round(Line *lines, uint64_t n, uint64_t round, int mode, unsigned long long *out) {
    const bool writer = (blockIdx.x & 1u) == 0u;
    const uint64_t pairs = (uint64_t)(gridDim.x / 2) * kBlock;
    const uint64_t t = (uint64_t)(blockIdx.x / 2) * kBlock + threadIdx.x;
    unsigned long long seen = 0, torn = 0, gave_up = 0;
    for (uint64_t i = t; i < n; i += pairs) {
        Line *L = lines + i;
        const unsigned long long prev = flag_of(round - 1, i) | kReady, claim = flag_of(round, i), ready = claim | kReady;
        const unsigned long long pay = payload_of(round, i);
        if (writer) {
            __builtin_amdgcn_s_sleep(20);                         // the reader gets there first and watches the line change
            const unsigned long long old = atomicCAS(&L->w[0], prev, claim);
            if (old != prev) { gave_up++; continue; }
            if (mode == 2) {
                store16(L, u32x4{(uint32_t)ready, (uint32_t)(ready >> 32), (uint32_t)pay, (uint32_t)(pay >> 32)});
            } else {
                st_agent(&L->w[mode == 3 ? 2 : 1], pay);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                st_agent(&L->w[0], ready);
            }
        } else {
            bool done = false;
            for (int spin = 0; spin < (1 << 16) && !done; ++spin) {
                const u32x4 a = load16(L);
                const unsigned long long f = (unsigned long long)a.x | ((unsigned long long)a.y << 32);
                if (f == ready) {
                    unsigned long long p = (unsigned long long)a.z | ((unsigned long long)a.w << 32);
                    if (mode == 3) { const u32x4 b = load16(&L->w[2]); p = (unsigned long long)b.x | ((unsigned long long)b.y << 32); }
                    if (spin > 0) { seen++; torn += p != pay; }
                    done = true;
                }
            }
            gave_up += !done;
        }
    }
    if (seen) atomicAdd(&out[0], seen);
    if (torn) atomicAdd(&out[1], torn);
    if (gave_up) atomicAdd(&out[writer ? 3 : 2], gave_up);
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 22;
    const int rounds = argc > 2 ? atoi(argv[2]) : 40;
    const uint64_t n = 1ull << lg;
    Line *lines = nullptr;
    unsigned long long *out = nullptr;
    CHECK(hipMalloc(&lines, n * sizeof(Line)));
    CHECK(hipMalloc(&out, 4 * sizeof(unsigned long long)));
    const int grid = 512;                     // two workgroups a CU: every workgroup is resident, writers never wait for anybody
    const char *what[4] = {"two 8-byte stores, wait between", "the same + release fence", "one 16-byte store", "payload in the next 16 bytes"};
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k_init, dim3(2048), dim3(kBlock), 0, 0, lines, n);
        CHECK(hipMemset(out, 0, 4 * sizeof(unsigned long long)));
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        CHECK(hipEventRecord(a, 0));
        for (int r = 1; r <= rounds; ++r) hipLaunchKernelGGL(k_round, dim3(grid), dim3(kBlock), 0, 0, lines, n, (uint64_t)r, mode, out);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        unsigned long long h[4];
        CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        printf("mode %d (%s): %llu line writes, %llu transitions watched, %llu TORN (READY with the old payload), readers gave up %llu, writers lost the CAS %llu, %.1f ms\n",
               mode, what[mode], (unsigned long long)(n * rounds), h[0], h[1], h[2], h[3], ms);
        fflush(stdout);
    }
    return 0;
}
