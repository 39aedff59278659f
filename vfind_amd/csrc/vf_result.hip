// vf_result.hip -- the (sequence, count) frame of src/lib.rs:312-319 built ON THE DEVICE from an exported table:
// key strings expanded, duplicate strings summed, rows ordered by count descending then sequence ascending, and only
// the three finished Arrow buffers (key bytes, int64 offsets, uint64 counts) cross PCIe.
//
// Round 1 copied the whole table to the host and rebuilt every key as a std::string in an unordered_map before a
// std::sort: seconds at the 10^7 keys of BASELINE configs[4].  Here:
//   k_rs_prep     per exported entry: string length, String::from_utf8 guard of lib.rs:295 (untranslated keys only)
//   scan          byte offsets of the kept entries, their compacted positions                       (rocPRIM scan)
//   k_rs_expand   key bytes (peptide letters as stored; 2-bit packed DNA expanded to ACGT), 8-byte big-endian prefix
//   sort 1        entries by string, ascending: merge sort on {prefix, position}; the comparator decides on the prefix
//                 in registers and reads key bytes only for equal prefixes                      (rocPRIM merge_sort)
//   k_rs_heads    first entry of every run of equal strings (duplicates: a stale table word on the device, or the
//                 same key exported by several GPUs before a merge); scan -> group ids; k_rs_groups sums the counts
//   sort 2        groups by (count descending, string rank ascending): a total order, no reliance on stability
//   k_rs_gather   offsets (scan of the lengths in final order), bytes and counts into the output buffers
// The sorts are rocPRIM's device primitives; everything that touches keys is written here.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vfind_hip.h"
#include "vf_device.h"
#include "vf_result.h"

#include <rocprim/device/device_merge_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace vf {

namespace {

constexpr int kRsBlock = 256;

struct Ent {                 // one kept entry
    uint64_t off;            // first byte of its string in the expanded key bytes
    uint64_t count;
    uint32_t len;            // bytes
    uint32_t pad;
};
struct SortKey {
    uint64_t prefix;         // first 8 bytes of the string, big-endian, zero padded: integer order = byte order
    uint32_t pos;            // index into Ent[]
    uint32_t len;
};

__device__ __forceinline__ bool utf8_ok(const uint8_t *s, uint32_t n) {
    uint32_t i = 0;
    while (i < n) {
        const uint8_t ch = s[i];
        if (ch < 0x80) { ++i; continue; }
        uint32_t len, cp;
        if ((ch & 0xE0) == 0xC0) { len = 2; cp = ch & 0x1F; }
        else if ((ch & 0xF0) == 0xE0) { len = 3; cp = ch & 0x0F; }
        else if ((ch & 0xF8) == 0xF0) { len = 4; cp = ch & 0x07; }
        else return false;
        if (i + len > n) return false;
        for (uint32_t k = 1; k < len; ++k) { if ((s[i + k] & 0xC0) != 0x80) return false; cp = (cp << 6) | (s[i + k] & 0x3F); }
        if ((len == 2 && cp < 0x80) || (len == 3 && cp < 0x800) || (len == 4 && (cp < 0x10000 || cp > 0x10FFFF))) return false;
        if (cp >= 0xD800 && cp <= 0xDFFF) return false;
        i += len;
    }
    return true;
}

// ref = key_off_dwords << 24 | valid << 23 | kind << 16 | len   (vf_device.h)
__global__ __launch_bounds__(kRsBlock) void k_rs_prep(const uint64_t *__restrict__ refs, const uint32_t *__restrict__ arena, uint64_t n,
                                                     int check_utf8, uint64_t *__restrict__ blen, uint32_t *__restrict__ keep) {
    const uint64_t i = (uint64_t)blockIdx.x * kRsBlock + threadIdx.x;
    if (i >= n) return;
    const uint64_t ref = refs[i];
    const uint32_t len = (uint32_t)(ref & 0xFFFFull), kind = (uint32_t)((ref >> 16) & 3ull);
    bool ok = true;
    // String::from_utf8 (lib.rs:295): only untranslated byte keys can fail it (peptides and ACGT strings are ASCII)
    if (check_utf8 && kind == kKindBytes) ok = utf8_ok(reinterpret_cast<const uint8_t *>(arena + (ref >> 24)), len);
    keep[i] = ok ? 1u : 0u;
    blen[i] = ok ? (uint64_t)len : 0ull;
}

__global__ __launch_bounds__(kRsBlock) void k_rs_expand(const uint64_t *__restrict__ refs, const uint64_t *__restrict__ counts,
                                                       const uint32_t *__restrict__ arena, uint64_t n, const uint64_t *__restrict__ boff,
                                                       const uint32_t *__restrict__ keep, const uint32_t *__restrict__ kpos,
                                                       uint8_t *__restrict__ bytes, Ent *__restrict__ ent, SortKey *__restrict__ sk) {
    const uint64_t i = (uint64_t)blockIdx.x * kRsBlock + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const uint64_t ref = refs[i];
    const uint32_t len = (uint32_t)(ref & 0xFFFFull), kind = (uint32_t)((ref >> 16) & 3ull);
    const uint32_t *k = arena + (ref >> 24);
    uint8_t *dst = bytes + boff[i];
    uint64_t prefix = 0;
    if (kind == kKindPacked) {
        for (uint32_t b = 0; b < len; ++b) {
            const uint8_t c = (uint8_t)"ACGT"[(k[b >> 4] >> (2u * (b & 15u))) & 3u];
            dst[b] = c;
            if (b < 8) prefix |= (uint64_t)c << (8u * (7u - b));
        }
    } else {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(k);
        for (uint32_t b = 0; b < len; ++b) {
            const uint8_t c = src[b];
            dst[b] = c;
            if (b < 8) prefix |= (uint64_t)c << (8u * (7u - b));
        }
    }
    const uint32_t p = kpos[i];
    ent[p] = Ent{boff[i], counts[i], len, 0u};
    sk[p] = SortKey{prefix, p, len};
}

// byte order of two strings whose first 8 bytes are equal: the rest of the bytes, then the shorter one first
__device__ __forceinline__ int cmp_tail(const uint8_t *__restrict__ bytes, const Ent &a, const Ent &b) {
    const uint32_t m = a.len < b.len ? a.len : b.len;
    const uint8_t *pa = bytes + a.off, *pb = bytes + b.off;
    for (uint32_t i = 8; i < m; ++i) {
        const int d = (int)pa[i] - (int)pb[i];
        if (d) return d;
    }
    return a.len < b.len ? -1 : (a.len > b.len ? 1 : 0);
}
struct StringLess {
    const uint8_t *bytes;
    const Ent *ent;
    __device__ bool operator()(const SortKey &a, const SortKey &b) const {
        if (a.prefix != b.prefix) return a.prefix < b.prefix;
        if (a.len <= 8 || b.len <= 8) {                    // (a string of <= 8 bytes is its prefix; NUL bytes: the length decides)
            if (a.len != b.len && (a.len <= 8 && b.len <= 8)) return a.len < b.len;
            if (a.len <= 8 && b.len > 8) return true;
            if (b.len <= 8 && a.len > 8) return false;
            return a.pos < b.pos;                          // same string: any fixed order
        }
        const int c = cmp_tail(bytes, ent[a.pos], ent[b.pos]);
        return c != 0 ? c < 0 : a.pos < b.pos;
    }
};
__device__ __forceinline__ bool same_string(const uint8_t *__restrict__ bytes, const Ent *__restrict__ ent, const SortKey &a, const SortKey &b) {
    if (a.prefix != b.prefix || a.len != b.len) return false;
    if (a.len <= 8) return true;
    return cmp_tail(bytes, ent[a.pos], ent[b.pos]) == 0;
}

__global__ __launch_bounds__(kRsBlock) void k_rs_heads(const SortKey *__restrict__ sk, const uint8_t *__restrict__ bytes,
                                                      const Ent *__restrict__ ent, uint64_t n, uint32_t *__restrict__ head) {
    const uint64_t i = (uint64_t)blockIdx.x * kRsBlock + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || !same_string(bytes, ent, sk[i], sk[i - 1])) ? 1u : 0u;
}
// gid = inclusive scan of head (1-based group of every sorted entry)
__global__ __launch_bounds__(kRsBlock) void k_rs_groups(const SortKey *__restrict__ sk, const Ent *__restrict__ ent, const uint32_t *__restrict__ head,
                                                       const uint32_t *__restrict__ gid, uint64_t n, unsigned long long *__restrict__ gcount,
                                                       uint32_t *__restrict__ grep) {
    const uint64_t i = (uint64_t)blockIdx.x * kRsBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = gid[i] - 1u;
    const Ent e = ent[sk[i].pos];
    if (head[i]) { grep[g] = sk[i].pos; if (i + 1 == n || head[i + 1]) { gcount[g] = e.count; return; } }
    atomicAdd(&gcount[g], (unsigned long long)e.count);      // a run of duplicates: rare
}
struct GroupLess {             // count descending, then string rank (= group index) ascending
    const unsigned long long *gcount;
    __device__ bool operator()(uint32_t a, uint32_t b) const {
        const unsigned long long ca = gcount[a], cb = gcount[b];
        return ca != cb ? ca > cb : a < b;
    }
};
__global__ __launch_bounds__(kRsBlock) void k_rs_iota(uint32_t *__restrict__ v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * kRsBlock + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
__global__ __launch_bounds__(kRsBlock) void k_rs_outlen(const uint32_t *__restrict__ order, const uint32_t *__restrict__ grep, const Ent *__restrict__ ent,
                                                       uint64_t g, uint64_t *__restrict__ olen) {
    const uint64_t j = (uint64_t)blockIdx.x * kRsBlock + threadIdx.x;
    if (j < g) olen[j] = ent[grep[order[j]]].len;
}
__global__ __launch_bounds__(kRsBlock) void k_rs_gather(const uint32_t *__restrict__ order, const uint32_t *__restrict__ grep, const Ent *__restrict__ ent,
                                                       const unsigned long long *__restrict__ gcount, const uint64_t *__restrict__ ooff,
                                                       const uint8_t *__restrict__ bytes, uint64_t g, uint8_t *__restrict__ out_keys,
                                                       uint64_t *__restrict__ out_offsets, uint64_t *__restrict__ out_counts) {
    const uint64_t j = (uint64_t)blockIdx.x * kRsBlock + threadIdx.x;
    if (j >= g) return;
    const uint32_t grp = order[j];
    const Ent e = ent[grep[grp]];
    const uint64_t o = ooff[j];
    for (uint32_t b = 0; b < e.len; ++b) out_keys[o + b] = bytes[e.off + b];
    out_offsets[j] = o;
    if (j + 1 == g) out_offsets[g] = o + e.len;
    out_counts[j] = gcount[grp];
}

struct Dev {                   // scoped device allocation
    void *p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

#define RS_TRY(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(_e); return VF_EDEVICE; } \
    } while (0)

inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + kRsBlock - 1) / kRsBlock); }

__global__ __launch_bounds__(kRsBlock) void k_rs_rebase(const uint64_t *__restrict__ refs, uint64_t n, uint64_t arena_base, uint64_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * kRsBlock + threadIdx.x;
    if (i < n) out[i] = refs[i] + (arena_base << 24);
}

}  // namespace

// the entry references of one exported table moved to a key arena that starts arena_base dwords further on (several
// exported tables laid end to end form one)
hipError_t rebase_refs(hipStream_t s, const uint64_t *d_refs, uint64_t n, uint64_t arena_base, uint64_t *d_out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rs_rebase, dim3(blocks_for(n)), dim3(kRsBlock), 0, s, d_refs, n, arena_base, d_out);
    return hipGetLastError();
}

// refs / counts / arena: an exported table in device memory (vf_table_export's blob); fills `out` (host vectors)
int build_result_on_device(hipStream_t s, const uint64_t *d_refs, const uint64_t *d_counts, const uint32_t *d_arena, uint64_t n,
                           bool check_utf8, vf_result &out, std::string &err) {
    out.keys.clear(); out.offsets.assign(1, 0); out.counts.clear(); out.dup_rows = 0;
    if (n == 0) return VF_OK;
    static const bool timing = getenv("VFIND_RESULT_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[result] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (n >= 0xFFFFFFF0ull) { err = "table too large for the device result build"; return VF_ECAPACITY; }
    Dev blen, keep, boff, kpos, tmp;
    RS_TRY(blen.alloc(n * 8)); RS_TRY(keep.alloc(n * 4)); RS_TRY(boff.alloc((n + 1) * 8)); RS_TRY(kpos.alloc((n + 1) * 4));
    hipLaunchKernelGGL(k_rs_prep, dim3(blocks_for(n)), dim3(kRsBlock), 0, s, d_refs, d_arena, n, check_utf8 ? 1 : 0, blen.as<uint64_t>(), keep.as<uint32_t>());
    RS_TRY(hipGetLastError());
    // exclusive scans: byte offsets and compacted positions (n + 1 outputs would need n + 1 inputs: totals come from the last pair)
    size_t tb = 0, tb2 = 0, tb3 = 0;
    RS_TRY(rocprim::exclusive_scan(nullptr, tb, blen.as<uint64_t>(), boff.as<uint64_t>(), (uint64_t)0, n, rocprim::plus<uint64_t>(), s));
    RS_TRY(rocprim::exclusive_scan(nullptr, tb2, keep.as<uint32_t>(), kpos.as<uint32_t>(), (uint32_t)0, n, rocprim::plus<uint32_t>(), s));
    RS_TRY(tmp.alloc(std::max(tb, tb2)));
    RS_TRY(rocprim::exclusive_scan(tmp.p, tb, blen.as<uint64_t>(), boff.as<uint64_t>(), (uint64_t)0, n, rocprim::plus<uint64_t>(), s));
    RS_TRY(rocprim::exclusive_scan(tmp.p, tb2, keep.as<uint32_t>(), kpos.as<uint32_t>(), (uint32_t)0, n, rocprim::plus<uint32_t>(), s));
    uint64_t last_off = 0, last_len = 0;
    uint32_t last_pos = 0, last_keep = 0;
    RS_TRY(hipMemcpyAsync(&last_off, boff.as<uint64_t>() + (n - 1), 8, hipMemcpyDeviceToHost, s));
    RS_TRY(hipMemcpyAsync(&last_len, blen.as<uint64_t>() + (n - 1), 8, hipMemcpyDeviceToHost, s));
    RS_TRY(hipMemcpyAsync(&last_pos, kpos.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, s));
    RS_TRY(hipMemcpyAsync(&last_keep, keep.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, s));
    RS_TRY(hipStreamSynchronize(s));
    const uint64_t total_bytes = last_off + last_len, m = (uint64_t)last_pos + last_keep;       // kept entries
    lap("prep + scans");
    if (m == 0) return VF_OK;

    Dev bytes, ent, sk, sk2, head, gid;
    RS_TRY(bytes.alloc(total_bytes)); RS_TRY(ent.alloc(m * sizeof(Ent))); RS_TRY(sk.alloc(m * sizeof(SortKey))); RS_TRY(sk2.alloc(m * sizeof(SortKey)));
    hipLaunchKernelGGL(k_rs_expand, dim3(blocks_for(n)), dim3(kRsBlock), 0, s, d_refs, d_counts, d_arena, n, boff.as<uint64_t>(), keep.as<uint32_t>(),
                       kpos.as<uint32_t>(), bytes.as<uint8_t>(), ent.as<Ent>(), sk.as<SortKey>());
    RS_TRY(hipGetLastError());
    lap("expand");
    // sort 1: by string
    const StringLess less{bytes.as<uint8_t>(), ent.as<Ent>()};
    Dev t1;
    RS_TRY(rocprim::merge_sort(nullptr, tb3, sk.as<SortKey>(), sk2.as<SortKey>(), (size_t)m, less, s));
    RS_TRY(t1.alloc(tb3));
    RS_TRY(rocprim::merge_sort(t1.p, tb3, sk.as<SortKey>(), sk2.as<SortKey>(), (size_t)m, less, s));
    lap("sort by string");
    // runs of equal strings -> groups
    RS_TRY(head.alloc(m * 4)); RS_TRY(gid.alloc(m * 4));
    hipLaunchKernelGGL(k_rs_heads, dim3(blocks_for(m)), dim3(kRsBlock), 0, s, sk2.as<SortKey>(), bytes.as<uint8_t>(), ent.as<Ent>(), m, head.as<uint32_t>());
    RS_TRY(hipGetLastError());
    size_t tb4 = 0;
    Dev t2;
    RS_TRY(rocprim::inclusive_scan(nullptr, tb4, head.as<uint32_t>(), gid.as<uint32_t>(), (size_t)m, rocprim::plus<uint32_t>(), s));
    RS_TRY(t2.alloc(tb4));
    RS_TRY(rocprim::inclusive_scan(t2.p, tb4, head.as<uint32_t>(), gid.as<uint32_t>(), (size_t)m, rocprim::plus<uint32_t>(), s));
    uint32_t g32 = 0;
    RS_TRY(hipMemcpyAsync(&g32, gid.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, s));
    RS_TRY(hipStreamSynchronize(s));
    const uint64_t g = g32;
    out.dup_rows = m - g;                                   // kept entries that shared their string with another one
    Dev gcount, grep, order, order2;
    RS_TRY(gcount.alloc(g * 8)); RS_TRY(grep.alloc(g * 4)); RS_TRY(order.alloc(g * 4)); RS_TRY(order2.alloc(g * 4));
    RS_TRY(hipMemsetAsync(gcount.p, 0, g * 8, s));
    hipLaunchKernelGGL(k_rs_groups, dim3(blocks_for(m)), dim3(kRsBlock), 0, s, sk2.as<SortKey>(), ent.as<Ent>(), head.as<uint32_t>(), gid.as<uint32_t>(), m,
                       gcount.as<unsigned long long>(), grep.as<uint32_t>());
    RS_TRY(hipGetLastError());
    lap("heads + groups");
    // sort 2: groups by (count descending, string ascending)
    hipLaunchKernelGGL(k_rs_iota, dim3(blocks_for(g)), dim3(kRsBlock), 0, s, order.as<uint32_t>(), g);
    RS_TRY(hipGetLastError());
    const GroupLess gless{gcount.as<unsigned long long>()};
    size_t tb5 = 0;
    Dev t3;
    RS_TRY(rocprim::merge_sort(nullptr, tb5, order.as<uint32_t>(), order2.as<uint32_t>(), (size_t)g, gless, s));
    RS_TRY(t3.alloc(tb5));
    RS_TRY(rocprim::merge_sort(t3.p, tb5, order.as<uint32_t>(), order2.as<uint32_t>(), (size_t)g, gless, s));
    lap("sort by count");
    // output buffers
    Dev olen, ooff, okeys, ooffs, ocnts, t4;
    RS_TRY(olen.alloc(g * 8)); RS_TRY(ooff.alloc(g * 8));
    hipLaunchKernelGGL(k_rs_outlen, dim3(blocks_for(g)), dim3(kRsBlock), 0, s, order2.as<uint32_t>(), grep.as<uint32_t>(), ent.as<Ent>(), g, olen.as<uint64_t>());
    RS_TRY(hipGetLastError());
    size_t tb6 = 0;
    RS_TRY(rocprim::exclusive_scan(nullptr, tb6, olen.as<uint64_t>(), ooff.as<uint64_t>(), (uint64_t)0, (size_t)g, rocprim::plus<uint64_t>(), s));
    RS_TRY(t4.alloc(tb6));
    RS_TRY(rocprim::exclusive_scan(t4.p, tb6, olen.as<uint64_t>(), ooff.as<uint64_t>(), (uint64_t)0, (size_t)g, rocprim::plus<uint64_t>(), s));
    RS_TRY(okeys.alloc(total_bytes)); RS_TRY(ooffs.alloc((g + 1) * 8)); RS_TRY(ocnts.alloc(g * 8));
    hipLaunchKernelGGL(k_rs_gather, dim3(blocks_for(g)), dim3(kRsBlock), 0, s, order2.as<uint32_t>(), grep.as<uint32_t>(), ent.as<Ent>(),
                       gcount.as<unsigned long long>(), ooff.as<uint64_t>(), bytes.as<uint8_t>(), g, okeys.as<uint8_t>(), ooffs.as<uint64_t>(),
                       ocnts.as<uint64_t>());
    RS_TRY(hipGetLastError());
    lap("gather");
    uint64_t out_bytes = 0;                                 // duplicates merged: at most total_bytes
    RS_TRY(hipMemcpyAsync(&out_bytes, ooffs.as<uint64_t>() + g, 8, hipMemcpyDeviceToHost, s));
    RS_TRY(hipStreamSynchronize(s));                        // s does not synchronise with the null stream the slice copies below use
    if (out_bytes > total_bytes) { err = "result build: inconsistent key bytes"; return VF_EDEVICE; }
    out.offsets.resize(g + 1);
    out.counts.resize(g);
    out.keys.resize(out_bytes);
    // the three buffers come back in slices copied by several host threads at once: the destination is ordinary (pageable,
    // freshly mapped) memory, so a copy is a DMA into the runtime's staging plus a CPU copy that also takes the page faults,
    // and one thread doing all of it was 110 ms of a 140 ms vf_finish at 10^7 rows / 470 MB of strings
    struct Piece { void *dst; const void *src; size_t n; };
    std::vector<Piece> pieces;
    auto add = [&](void *dst, const void *src, size_t nbytes) {
        const size_t slice = 32u << 20;
        for (size_t at = 0; at < nbytes; at += slice) pieces.push_back(Piece{(uint8_t *)dst + at, (const uint8_t *)src + at, std::min(slice, nbytes - at)});
    };
    add(out.offsets.data(), ooffs.p, (g + 1) * 8);
    add(out.counts.data(), ocnts.p, g * 8);
    add(out.keys.data(), okeys.p, out_bytes);
    int dev = 0;
    RS_TRY(hipGetDevice(&dev));
    const unsigned nthreads = (unsigned)std::min<size_t>(8, pieces.size());
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&] {
        if (hipSetDevice(dev) != hipSuccess) { bad = 1; return; }
        for (size_t i = next++; i < pieces.size(); i = next++)
            if (hipMemcpy(pieces[i].dst, pieces[i].src, pieces[i].n, hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
    };
    std::vector<std::thread> th;
    for (unsigned k = 1; k < nthreads; ++k) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (bad) { err = "copy of the result frame failed"; return VF_EDEVICE; }
    lap("copy back");
    return VF_OK;
}

}  // namespace vf
