// CPU harness for the gzip FASTQ reader (vfind_amd/csrc/vf_fastq.cpp): stubs the device submission
// path and prints the number of reads and an order-independent checksum of the sequences.
// Build: g++ -O2 -std=c++17 -I include tests/cpp/fastq_harness.cpp vfind_amd/csrc/vf_fastq.cpp -lz -lpthread
#include "../../include/vfind_hip.h"
#include "../../vfind_amd/csrc/vf_internal.h"

#include <cstring>
#include <cstdlib>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>

struct vf_ctx { uint32_t threads; };
static std::atomic<uint64_t> g_reads{0}, g_sum{0}, g_bases{0};
static std::mutex g_mu;
static int g_rc = 0;
static std::string g_msg;
static thread_local std::string t_err;

int vf_internal_submit(vf_ctx *, const uint8_t *base, const ReadRef *reads, uint64_t n) {
    uint64_t s = 0, b = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t h = 1469598103934665603ull;
        for (uint32_t k = 0; k < reads[i].len; ++k) { h ^= base[reads[i].off + k]; h *= 1099511628211ull; }
        s += h;
        b += reads[i].len;
    }
    g_reads += n; g_sum += s; g_bases += b;
    return 0;
}
// device framing is a GPU path; the harness exercises the host framer
static VfFastqStats g_fq_stats;
VfFastqStats *vf_internal_fastq_stats(vf_ctx *) { return &g_fq_stats; }
bool vf_internal_device_framing(const vf_ctx *) { return false; }
uint8_t *vf_internal_alloc_pinned(const vf_ctx *, size_t n) { return (uint8_t *)malloc(n); }
void vf_internal_free_pinned(uint8_t *p) { free(p); }
int vf_internal_submit_text(vf_ctx *, const TextSpan *, size_t, bool, bool) { t_err = "device framing is not available in the CPU harness"; return 4; }
void vf_internal_set_async_error(vf_ctx *, int rc, const std::string &msg) {
    std::lock_guard<std::mutex> g(g_mu);
    if (!g_rc) { g_rc = rc; g_msg = msg; }
}
uint32_t vf_internal_threads(const vf_ctx *c) { return c->threads; }
int vf_internal_fail(int code, const std::string &msg) { t_err = msg; return code; }
std::string vf_internal_last_error() { return t_err; }
int vf_internal_fail_noexcept(int code, const char *msg) noexcept { try { t_err = msg; } catch (...) { } return code; }
bool vf_internal_test_throw(const char *where) { const char *e = getenv("VFIND_TEST_THROW"); return e && strcmp(e, where) == 0; }
extern "C" int vf_sync(vf_ctx *) {
    std::lock_guard<std::mutex> g(g_mu);
    if (g_rc) { t_err = g_msg; return g_rc; }
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s file.fq.gz n_threads\n", argv[0]); return 2; }
    vf_ctx c{(uint32_t)atoi(argv[2])};
    const int rc = vf_run_fastq(&c, argv[1]);
    printf("{\"rc\": %d, \"reads\": %llu, \"bases\": %llu, \"checksum\": %llu, \"error\": \"%s\"}\n", rc, (unsigned long long)g_reads.load(),
           (unsigned long long)g_bases.load(), (unsigned long long)g_sum.load(), rc ? t_err.c_str() : "");
    return 0;
}
