// vf_internal.h -- glue between the host runtime (vf_host.cpp) and the FASTQ reader (vf_fastq.cpp)
#pragma once
#include <stdint.h>

#include <memory>
#include <new>
#include <exception>
#include <string>
#include <utility>
#include <vector>

struct vf_ctx;
struct ReadRef { uint64_t off; uint32_t len; };   // a sequence inside a text buffer

int vf_internal_submit(vf_ctx *c, const uint8_t *base, const ReadRef *reads, uint64_t n);
// a chunk of whole FASTQ records as byte ranges; framing and packing happen on the device
struct TextSpan { const uint8_t *p; size_t n; };
int vf_internal_submit_text(vf_ctx *c, const TextSpan *spans, size_t n_spans, bool final_chunk, bool spans_pinned);
// whether text chunks are framed and packed on the device (default) or on the reader's threads
bool vf_internal_device_framing(const vf_ctx *c);
// page-locked host memory for the inflate pieces, so that text goes to the device straight from them
uint8_t *vf_internal_alloc_pinned(const vf_ctx *c, size_t bytes);
void vf_internal_free_pinned(uint8_t *p);
void vf_internal_set_async_error(vf_ctx *c, int rc, const std::string &msg);
uint32_t vf_internal_threads(const vf_ctx *c);
int vf_internal_fail(int code, const std::string &msg);      // sets the calling thread's error text
std::string vf_internal_last_error();
int vf_internal_fail_noexcept(int code, const char *msg) noexcept;   // the same for a message that must not allocate

// stage timers of vf_run_fastq (vf_stats.fq_*): nanoseconds / bytes, added to by the reader's threads
#include <atomic>
struct VfFastqStats {
    std::atomic<uint64_t> wall_ns{0}, inflate_ns{0}, seq_wait_ns{0}, queue_full_ns{0}, submit_ns{0}, slot_wait_ns{0}, index_wait_ns{0}, drain_ns{0};
    std::atomic<uint64_t> comp_bytes{0}, text_bytes{0}, threads{0}, chunks{0};
    void reset() { for (auto *a : {&wall_ns, &inflate_ns, &seq_wait_ns, &queue_full_ns, &submit_ns, &slot_wait_ns, &index_wait_ns, &drain_ns, &comp_bytes, &text_bytes, &threads, &chunks}) a->store(0); }
};
VfFastqStats *vf_internal_fastq_stats(vf_ctx *c);

// The exception barrier of the C ABI (include/vfind_hip.h: "no exceptions or panics cross the ABI"): every extern "C" entry
// point runs its body inside vf_guard, so a std::bad_alloc or a length_error from a vector sized by a device-read value
// comes back as an error code and a message, never as std::terminate in the host process - what lib.rs:317 does with a
// failed frame build (PyValueError) and what lib.rs:155, 308 do not (unwrap -> panic).
template <class F> static inline int vf_guard(F &&f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc &) { return vf_internal_fail_noexcept(6 /* VF_ENOMEM */, "out of host memory"); }
    catch (const std::exception &e) {
        try { return vf_internal_fail(4 /* VF_EDEVICE */, std::string("internal error: ") + e.what()); }
        catch (...) { return vf_internal_fail_noexcept(4, "internal error"); }
    }
    catch (...) { return vf_internal_fail_noexcept(4, "internal error: unknown exception"); }
}
// test hook: VFIND_TEST_THROW=<where> makes the named place throw (tests/test_gpu_parity.py::test_exception_barrier...)
bool vf_internal_test_throw(const char *where);

// the (sequence, count) frame handed over the ABI (lib.rs:312-319): rows sorted count desc, sequence asc
// (vector storage that resize() leaves uninitialised: the buffers are filled by a device-to-host copy, and zeroing 600 MB
// first cost more than the copy)
template <class T> struct VfNoInit : std::allocator<T> {
    template <class U> struct rebind { using other = VfNoInit<U>; };
    VfNoInit() = default;
    template <class U> VfNoInit(const VfNoInit<U> &) {}
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
struct vf_result {
    std::vector<uint8_t, VfNoInit<uint8_t>> keys;        // concatenated UTF-8 keys
    std::vector<uint64_t, VfNoInit<uint64_t>> offsets;   // rows + 1
    std::vector<uint64_t, VfNoInit<uint64_t>> counts;    // rows
    uint64_t dup_rows = 0;                               // entries folded into another row (equal strings), vf_stats.dup_rows
};

