// vf_fastq.cpp -- gzip FASTQ input for vf_run_fastq (replaces File::open -> MultiGzDecoder -> Reader
// -> parallel_fastq, src/lib.rs:233-234, 271-274).
//
//   [inflate pool] --text pieces--> [sequencer] --whole-record chunks--> [submit pool] --> GPU
//   (the device indexes lines, frames records and packs; with VFIND_FASTQ_DEVICE_FRAMING=0 the submit pool does)
//
// * The compressed file is mmapped.  gzip members are independent deflate streams, so members are
//   inflated in parallel (vf_inflate.cpp; VFIND_INFLATE=zlib selects zlib instead).  A decode task starts at a member start and runs through whole
//   members until the first member end at or beyond the next 4 MiB grid line that is itself a member
//   start (or EOF).  Task starts are claimed speculatively ("first gzip magic at or after the grid
//   line"): exact for BGZF and for ordinary multi-member files; a claim that was wrong (magic bytes
//   inside compressed data) is simply never consumed, and the sequencer requests the task it actually
//   needs, so correctness never depends on the speculation.  A file of few, large members (the usual
//   `gzip reads.fastq`) is handled the other way round: one task at a time, each member decoded by all
//   threads (vf_pinflate.cpp: cells of the stream decoded with an unknown window, then resolved).
// * The sequencer consumes tasks in file order (a task is consumed only if it starts exactly where the
//   previous one ended), counts newlines to cut the text into chunks of whole four-line records and
//   hands them to the frame+pack workers, which call the batch submission path.
// * Every member's CRC32/ISIZE trailer is verified; trailing garbage, truncation and malformed
//   records surface as VF_EFORMAT (the reference panics there, lib.rs:308).
#include "../../include/vfind_hip.h"
#include "vf_internal.h"
#include "vf_inflate.h"
#include "vf_pinflate.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <stdexcept>
#include <functional>
#include <cerrno>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr uint64_t kGrid = 4ull << 20;         // compressed bytes per decode task (at least)
constexpr size_t kPiece = 8u << 20;            // decoded text piece
static double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static size_t spec_cap() {                     // text a not-yet-needed task may buffer (tests shrink it)
    static const size_t v = [] { const char *e = getenv("VFIND_FASTQ_SPEC_CAP"); return e ? (size_t)strtoull(e, nullptr, 0) : (size_t)(96u << 20); }();
    return v;
}
constexpr size_t kChunkTarget = 16u << 20;     // text per frame+pack chunk
constexpr size_t kChunkCap = kChunkTarget + 2 * kPiece;   // its buffer (a chunk is cut once it passes the target)
constexpr size_t kTotalCap = 1024u << 20;       // decoded text all tasks together may hold
constexpr uint64_t kNone = ~0ull;

// Fixed-size byte buffers recycled through a free list: allocating and first-touching 8 MiB vectors per
// piece made the inflate threads serialise on page faults (6 parallel tasks each ran 5x slower).
struct BufPool {
    std::mutex mu;
    std::vector<uint8_t *> free_list;
    size_t bytes;
    const vf_ctx *ctx = nullptr;
    bool pinned = false;                       // page-locked buffers (device framing: text goes to the GPU straight from the pieces)
    explicit BufPool(size_t b) : bytes(b) {}
    std::vector<uint8_t *> locked;             // the page-locked ones (an allocation that fails falls back to malloc)
    bool is_locked(uint8_t *p) const { return std::find(locked.begin(), locked.end(), p) != locked.end(); }
    ~BufPool() { for (auto *p : free_list) { if (is_locked(p)) vf_internal_free_pinned(p); else free(p); } }
    uint8_t *get() {
        { std::lock_guard<std::mutex> g(mu); if (!free_list.empty()) { uint8_t *p = free_list.back(); free_list.pop_back(); return p; } }
        if (pinned) {
            uint8_t *p = vf_internal_alloc_pinned(ctx, bytes);
            if (p) { std::lock_guard<std::mutex> g(mu); locked.push_back(p); return p; }
        }
        return (uint8_t *)malloc(bytes);
    }
    void put(uint8_t *p) { if (p) { std::lock_guard<std::mutex> g(mu); free_list.push_back(p); } }
};

struct Piece { uint8_t *data = nullptr; size_t size = 0; size_t nl = 0; };      // nl = newlines, counted by the inflate worker

// newline count of a piece: the inflate workers run it in parallel while the text is still in their
// caches, so the one sequencer thread never scans text itself
static size_t count_newlines(const uint8_t *p, size_t n) { return vf::count_byte(p, n, '\n'); }

struct Task {
    uint64_t start = 0, end = 0;
    std::atomic<uint64_t> in_pos{0};           // absolute compressed offset consumed so far
    std::deque<Piece> out;
    size_t buffered = 0;
    bool done = false, failed = false, abort = false;
    std::string err;
};

// A chunk of whole records is a list of byte ranges inside inflate pieces; a piece goes back to its pool
// when the last chunk that references it has been copied out by a frame+pack worker.
struct PieceHold {
    uint8_t *data; BufPool *pool;
    PieceHold(uint8_t *d, BufPool *p) : data(d), pool(p) {}
    ~PieceHold() { pool->put(data); }
};
struct Seg { std::shared_ptr<PieceHold> hold; const uint8_t *p = nullptr; size_t n = 0; };
struct Chunk { std::vector<Seg> segs; size_t size = 0; bool final = false; };

template <typename T> struct BoundedQueue {
    std::mutex mu;
    std::condition_variable cv_put, cv_get;
    std::deque<T> q;
    size_t limit;
    bool closed = false;
    explicit BoundedQueue(size_t lim) : limit(lim) {}
    bool put(T v) {
        std::unique_lock<std::mutex> lk(mu);
        cv_put.wait(lk, [&] { return q.size() < limit || closed; });
        if (closed) return false;
        q.push_back(std::move(v));
        cv_get.notify_one();
        return true;
    }
    bool get(T &v) {
        std::unique_lock<std::mutex> lk(mu);
        cv_get.wait(lk, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        v = std::move(q.front());
        q.pop_front();
        cv_put.notify_one();
        return true;
    }
    void close(bool drop) {
        { std::lock_guard<std::mutex> g(mu); closed = true; if (drop) q.clear(); }
        cv_get.notify_all(); cv_put.notify_all();
    }
};

struct Reader {
    const uint8_t *file = nullptr;
    uint64_t size = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::map<uint64_t, std::shared_ptr<Task>> tasks;
    uint64_t head_pos = 0;                     // next task start the sequencer needs
    std::atomic<uint64_t> head_in{0};          // compressed offset the validated stream has consumed
    uint64_t required = kNone;                 // task the sequencer is waiting for and nobody decodes yet
    uint64_t spec_last = kNone;                // start of the latest claimed task (speculation frontier)
    size_t total_buffered = 0;                 // decoded text waiting in all tasks
    BufPool pieces{kPiece};
    bool stop = false;
    bool big_members = false;                  // few, large members: one task at a time, each member decoded by all threads
    uint32_t par_threads = 1;
    VfFastqStats *fq = nullptr;                // stage timers of the context (vf_stats.fq_*)

    bool is_member_start(uint64_t off) const {
        return off + 18 <= size && file[off] == 0x1f && file[off + 1] == 0x8b && file[off + 2] == 8 && (file[off + 3] & 0xE0) == 0;
    }
    uint64_t next_magic(uint64_t from) const {  // first plausible member start >= from, or kNone
        while (from + 18 <= size) {
            const void *p = memchr(file + from, 0x1f, size - from);
            if (!p) return kNone;
            const uint64_t off = (uint64_t)((const uint8_t *)p - file);
            if (is_member_start(off)) return off;
            from = off + 1;
        }
        return kNone;
    }
};

static uint64_t next_grid(uint64_t s) { return (s / kGrid + 1) * kGrid; }

// inflate whole members from t->start until the task's end rule fires
static void run_task_zlib(Reader &R, const std::shared_ptr<Task> &t);
static void run_task(Reader &R, const std::shared_ptr<Task> &t) {
    static const bool use_zlib = [] { const char *e = getenv("VFIND_INFLATE"); return e && !strcmp(e, "zlib"); }();
    if (use_zlib) { run_task_zlib(R, t); return; }
    std::unique_ptr<vf::GzipMemberDecoder> dec(new vf::GzipMemberDecoder);
    const uint8_t *file_end = R.file + R.size;
    const uint64_t stop_at = next_grid(t->start);
    dec->begin(R.file + t->start, file_end);
    Piece piece;
    piece.data = R.pieces.get();
    bool oom = piece.data == nullptr;
    size_t fill = 0;
    auto flush_piece = [&](bool final_piece) -> bool {      // returns false when aborted
        const size_t nl_count = fill ? count_newlines(piece.data, fill) : 0;
        std::unique_lock<std::mutex> lk(R.mu);
        if (fill) {
            piece.size = fill;
            piece.nl = nl_count;
            t->buffered += fill;
            R.total_buffered += fill;
            t->out.push_back(piece);
            piece = Piece();
            fill = 0;
            R.cv.notify_all();
        }
        if (!final_piece && !piece.data) {
            lk.unlock(); piece.data = R.pieces.get(); lk.lock();
            if (!piece.data) { oom = true; return false; }       // neither page-locked nor ordinary memory left
        }
        if (final_piece) return true;
        // a task nobody needs yet may only run so far ahead; and it gives way when the sequencer is
        // blocked on a task no worker is decoding (a false start can inflate garbage for a long time,
        // and with few workers it would otherwise hold the last one)
        auto starving = [&] { return R.required != kNone && R.required != t->start && !R.tasks.count(R.required); };
        R.cv.wait(lk, [&] { return t->abort || R.stop || t->start == R.head_pos || starving() || t->buffered <= spec_cap(); });
        if (t->start != R.head_pos && starving()) t->abort = true;
        return !(t->abort || R.stop);
    };
    bool ok = !oom;
    std::string err = oom ? "out of memory: no buffer for inflated text" : "";
    uint64_t end = 0;
    if (R.big_members && ok) {
        // one long deflate stream: every member is decoded by all threads (vf_pinflate.cpp)
        R.pieces.put(piece.data);
        piece = Piece();
        vf::ParallelInflateIO io;
        io.piece_bytes = kPiece;
        io.get_buffer = [&R]() { return R.pieces.get(); };
        io.drop_buffer = [&R](uint8_t *b) { R.pieces.put(b); };
        io.progress = [&t](uint64_t off) { t->in_pos.store(off, std::memory_order_relaxed); };
        uint64_t member = t->start;
        io.put_piece = [&](uint8_t *b, size_t n, size_t nl) -> bool {
            std::unique_lock<std::mutex> lk(R.mu);
            Piece pc; pc.data = b; pc.size = n; pc.nl = nl;
            R.fq->text_bytes += n;
            t->buffered += n;
            R.total_buffered += n;
            t->out.push_back(pc);
            R.cv.notify_all();
            // the sequencer drains this task as it goes; keep at most 512 MiB of text ahead of it
            R.cv.wait(lk, [&] { return t->abort || R.stop || t->buffered <= ((size_t)512 << 20); });
            return !(t->abort || R.stop);
        };
        for (;;) {
            const uint8_t *mend = nullptr;
            const char *perr = "";
            io.progress = [&t, member](uint64_t off) { t->in_pos.store(member + off, std::memory_order_relaxed); };
            const double td = now_s();
            const bool pok = vf::inflate_member_parallel(R.file + member, file_end, R.par_threads, io, &mend, &perr);
            R.fq->inflate_ns += (uint64_t)((now_s() - td) * 1e9) * R.par_threads;      // (wall time of a decode all threads work on)
            if (!pok) { ok = false; err = perr; break; }
            const uint64_t e = (uint64_t)(mend - R.file);
            t->in_pos.store(e, std::memory_order_relaxed);
            if (e >= R.size) { end = R.size; break; }
            if (e >= stop_at && R.is_member_start(e)) { end = e; break; }
            member = e;                                      // next member of the same task (or trailing garbage: header check)
        }
        std::lock_guard<std::mutex> g(R.mu);
        t->failed = !ok;
        t->err = err;
        t->end = end;
        t->done = true;
        R.cv.notify_all();
        return;
    }
    while (ok) {
        if (kPiece - fill < 4096 && !flush_piece(false)) { ok = false; err = oom ? "out of memory: no buffer for inflated text" : "aborted"; break; }
        size_t got = 0;
        const double td = now_s();
        const vf::GzipMemberDecoder::Status st = dec->run(piece.data + fill, kPiece - fill, &got);
        R.fq->inflate_ns += (uint64_t)((now_s() - td) * 1e9);
        R.fq->text_bytes += got;
        fill += got;
        t->in_pos.store((uint64_t)(dec->in_pos() - R.file), std::memory_order_relaxed);
        if (st == vf::GzipMemberDecoder::kError) { ok = false; err = dec->error(); break; }
        if (st == vf::GzipMemberDecoder::kPieceFull) {
            if (!flush_piece(false)) { ok = false; err = oom ? "out of memory: no buffer for inflated text" : "aborted"; break; }
            continue;
        }
        const uint64_t e = (uint64_t)(dec->in_pos() - R.file);      // absolute offset just past this member
        if (e >= R.size) { end = R.size; break; }
        if (e >= stop_at && R.is_member_start(e)) { end = e; break; }
        dec->begin(R.file + e, file_end);       // next member of the same task (or trailing garbage, which the header check rejects)
    }
    if (ok) (void)flush_piece(true);
    R.pieces.put(piece.data);
    std::lock_guard<std::mutex> g(R.mu);
    t->failed = !ok;
    t->err = err;
    t->end = end;
    t->done = true;
    R.cv.notify_all();
}

// the same task on zlib (VFIND_INFLATE=zlib): kept for A/B runs
static void run_task_zlib(Reader &R, const std::shared_ptr<Task> &t) {
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, 31) != Z_OK) {
        std::lock_guard<std::mutex> g(R.mu);
        t->failed = true; t->err = "zlib init failed"; t->done = true; R.cv.notify_all();
        return;
    }
    uint64_t pos = t->start;
    const uint64_t stop_at = next_grid(t->start);
    Piece piece;
    piece.data = R.pieces.get();
    bool oom = piece.data == nullptr;
    size_t fill = 0;
    auto flush_piece = [&](bool final_piece) -> bool {      // returns false when aborted
        const size_t nl_count = fill ? count_newlines(piece.data, fill) : 0;
        std::unique_lock<std::mutex> lk(R.mu);
        if (fill) {
            piece.size = fill;
            piece.nl = nl_count;
            t->buffered += fill;
            R.total_buffered += fill;
            t->out.push_back(piece);
            piece = Piece();
            fill = 0;
            R.cv.notify_all();
        }
        if (!final_piece && !piece.data) {
            lk.unlock(); piece.data = R.pieces.get(); lk.lock();
            if (!piece.data) { oom = true; return false; }       // neither page-locked nor ordinary memory left
        }
        if (final_piece) return true;
        // a task nobody needs yet may only run so far ahead; and it gives way when the sequencer is
        // blocked on a task no worker is decoding (a false start can inflate garbage for a long time,
        // and with few workers it would otherwise hold the last one)
        auto starving = [&] { return R.required != kNone && R.required != t->start && !R.tasks.count(R.required); };
        R.cv.wait(lk, [&] { return t->abort || R.stop || t->start == R.head_pos || starving() || t->buffered <= spec_cap(); });
        if (t->start != R.head_pos && starving()) t->abort = true;
        return !(t->abort || R.stop);
    };
    bool ok = !oom;
    std::string err = oom ? "out of memory: no buffer for inflated text" : "";
    uint64_t end = 0;
    while (ok) {
        if (z.avail_in == 0) {
            if (pos >= R.size) { ok = false; err = "gzip error: truncated stream"; break; }
            const uint64_t take = std::min<uint64_t>(R.size - pos, 1u << 30);
            z.next_in = const_cast<Bytef *>(R.file + pos);
            z.avail_in = (uInt)take;
            pos += take;
        }
        z.next_out = piece.data + fill;
        z.avail_out = (uInt)(kPiece - fill);
        const double td = now_s();
        const size_t fill0 = fill;
        const int rc = inflate(&z, Z_NO_FLUSH);
        fill = kPiece - z.avail_out;
        R.fq->inflate_ns += (uint64_t)((now_s() - td) * 1e9);
        R.fq->text_bytes += fill - fill0;
        t->in_pos.store(pos - z.avail_in, std::memory_order_relaxed);
        if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
            ok = false; err = std::string("gzip error: ") + (z.msg ? z.msg : "corrupt stream"); break;
        }
        if (fill == kPiece) { if (!flush_piece(false)) { ok = false; err = oom ? "out of memory: no buffer for inflated text" : "aborted"; break; } }
        if (rc == Z_STREAM_END) {
            const uint64_t e = pos - z.avail_in;            // absolute offset just past this member
            if (e >= R.size) { end = R.size; break; }
            if (e >= stop_at && R.is_member_start(e)) { end = e; break; }
            // next member of the same task (or trailing garbage, which inflate will reject)
            const uInt keep_in = z.avail_in;
            Bytef *keep_ptr = z.next_in;
            if (inflateReset(&z) != Z_OK) { ok = false; err = "zlib reset failed"; break; }
            z.next_in = keep_ptr;
            z.avail_in = keep_in;
        } else if (rc == Z_BUF_ERROR && z.avail_in == 0 && pos >= R.size) {
            ok = false; err = "gzip error: truncated stream"; break;
        }
    }
    inflateEnd(&z);
    if (ok) (void)flush_piece(true);
    R.pieces.put(piece.data);
    std::lock_guard<std::mutex> g(R.mu);
    t->failed = !ok;
    t->err = err;
    t->end = end;
    t->done = true;
    R.cv.notify_all();
}

static const bool g_dbg = getenv("VFIND_FASTQ_DEBUG") != nullptr;

static void inflate_worker(Reader &R) {
    for (;;) {
        std::shared_ptr<Task> t;
        {
            std::unique_lock<std::mutex> lk(R.mu);
            for (;;) {
                if (R.stop) return;
                uint64_t start = kNone;
                bool retry = false;
                if (R.required != kNone && !R.tasks.count(R.required)) {
                    start = R.required;                      // what the sequencer is blocked on
                    R.spec_last = start;                     // speculation restarts from a known-true boundary
                } else if (!R.big_members && R.spec_last != kNone && R.total_buffered < kTotalCap) {
                    // speculative: first member-like magic at or after the next grid line
                    const uint64_t from = next_grid(R.spec_last);
                    if (from < R.size) {
                        lk.unlock();
                        const uint64_t cand = R.next_magic(from);
                        lk.lock();
                        if (R.stop) return;
                        retry = true;          // the lock was dropped: re-evaluate everything (a request may have arrived)
                        if (next_grid(R.spec_last) != from) {}                     // another worker moved the frontier
                        else if (cand == kNone) R.spec_last = R.size;              // nothing more to guess
                        else if (R.tasks.count(cand) || cand < R.head_in.load()) R.spec_last = cand;
                        else start = cand;
                    }
                }
                if (start != kNone) {
                    t = std::make_shared<Task>();
                    t->start = start;
                    t->in_pos = start;
                    R.tasks[start] = t;
                    if (R.spec_last == kNone || start > R.spec_last) R.spec_last = start;
                    if (R.required == start) R.required = kNone;
                    break;
                }
                if (!retry) R.cv.wait(lk);
            }
        }
        const double t0 = now_s();
        run_task(R, t);
        {
            std::lock_guard<std::mutex> g(R.mu);
            if (t->abort || t->failed) {                    // abandoned or false start: drop what it buffered
                for (auto &pc : t->out) R.pieces.put(pc.data);
                R.total_buffered -= std::min(R.total_buffered, t->buffered);
                t->out.clear();
                t->buffered = 0;
                auto it = R.tasks.find(t->start);
                if (t->abort && it != R.tasks.end() && it->second == t && t->start != R.head_pos) R.tasks.erase(it);
            }
        }
        R.cv.notify_all();
        if (g_dbg) fprintf(stderr, "[fastq] task start=%llu end=%llu failed=%d abort=%d  %.3f..%.3f s\n", (unsigned long long)t->start,
                           (unsigned long long)t->end, (int)t->failed, (int)t->abort, t0, now_s());
    }
}

// frames strict four-line records in text[0..n); every record must be complete
static int frame_records(const uint8_t *t, size_t n, std::vector<ReadRef> &out, std::string &err) {
    size_t p = 0;
    auto line_end = [&](size_t from) -> size_t {
        const void *q = memchr(t + from, '\n', n - from);
        return q ? (size_t)((const uint8_t *)q - t) : n;
    };
    while (p < n) {
        const size_t e1 = line_end(p);
        if (e1 == p || (e1 == p + 1 && t[p] == '\r')) {            // blank line: only legal as trailing padding
            size_t k = p;
            while (k < n && (t[k] == '\n' || t[k] == '\r')) ++k;
            if (k == n) return VF_OK;
            err = "malformed FASTQ: blank line inside the file";
            return VF_EFORMAT;
        }
        if (t[p] != '@') { err = "malformed FASTQ: record does not start with '@'"; return VF_EFORMAT; }
        if (e1 >= n) { err = "malformed FASTQ: truncated record"; return VF_EFORMAT; }
        const size_t s2 = e1 + 1, e2 = line_end(s2);
        if (e2 >= n) { err = "malformed FASTQ: truncated record"; return VF_EFORMAT; }
        size_t sl = e2 - s2;
        if (sl && t[e2 - 1] == '\r') --sl;
        const size_t s3 = e2 + 1, e3 = line_end(s3);
        if (e3 >= n || s3 >= n || t[s3] != '+') { err = "malformed FASTQ: missing '+' separator line"; return VF_EFORMAT; }
        const size_t s4 = e3 + 1, e4 = line_end(s4);
        size_t ql = e4 - s4;
        if (ql && t[e4 - 1] == '\r') --ql;
        if (ql != sl) { err = "malformed FASTQ: sequence and quality lengths differ"; return VF_EFORMAT; }
        if (sl > 0xFFFFFFFFull) { err = "read too long"; return VF_EFORMAT; }
        out.push_back(ReadRef{(uint64_t)s2, (uint32_t)sl});
        p = e4 < n ? e4 + 1 : n;
    }
    return VF_OK;
}

}  // namespace

static int vf_run_fastq_impl(vf_ctx *c, const char *path);
extern "C" int vf_run_fastq(vf_ctx *c, const char *path) { return vf_guard([&] { return vf_run_fastq_impl(c, path); }); }
static int vf_run_fastq_impl(vf_ctx *c, const char *path) {
    if (!c || !path) return vf_internal_fail(VF_EINVAL, "null argument");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return vf_internal_fail(VF_EIO, std::string(path) + ": " + strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return vf_internal_fail(VF_EIO, std::string(path) + ": fstat failed"); }
    Reader R;
    R.size = (uint64_t)st.st_size;
    R.fq = vf_internal_fastq_stats(c);
    const double t_call = now_s();
    struct Wall { VfFastqStats *f; double t0; ~Wall() { f->wall_ns += (uint64_t)((now_s() - t0) * 1e9); } } wall{R.fq, t_call};
    R.fq->comp_bytes += R.size;
    void *map = nullptr;
    if (R.size) {
        map = mmap(nullptr, R.size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (map == MAP_FAILED) { close(fd); return vf_internal_fail(VF_EIO, std::string(path) + ": mmap failed"); }
        (void)madvise(map, R.size, MADV_SEQUENTIAL);
    }
    close(fd);
    R.file = (const uint8_t *)map;
    struct Unmap { void *p; uint64_t n; ~Unmap() { if (p) munmap(p, n); } } unmap{map, R.size};
    if (R.size < 2 || R.file[0] != 0x1f || R.file[1] != 0x8b)
        return vf_internal_fail(VF_EFORMAT, std::string(path) + ": not a gzip file (the reference reads gzip FASTQ only)");

    const uint32_t nthreads = std::max<uint32_t>(1, vf_internal_threads(c));
    {
        // Few large members (an ordinary `gzip file.fastq`) or many small ones (bgzip, pigz -i, concatenations)?
        // Member-level parallelism needs the latter; for the former each member is decoded by all threads.
        // (A false magic inside compressed data turns up about once per 134 MB and does not change the picture.)
        const uint64_t span = std::min<uint64_t>(R.size, 64ull << 20);
        uint64_t n_starts = 0, from = 1;
        while (from < span) { const uint64_t m = R.next_magic(from); if (m == kNone || m >= span) break; ++n_starts; from = m + 1; }
        const char *e = getenv("VFIND_INFLATE_PARALLEL");
        const int forced = e ? atoi(e) : -1;
        R.par_threads = nthreads;
        R.big_members = forced >= 0 ? forced != 0 : (nthreads >= 3 && R.size >= (8ull << 20) && span / (n_starts + 1) >= (24ull << 20));
    }
    BoundedQueue<std::unique_ptr<Chunk>> chunks(std::max<size_t>(2, nthreads));
    BufPool chunk_pool(kChunkCap);
    std::atomic<bool> failed{false};
    auto fail_async = [&](int rc, const std::string &msg) {
        vf_internal_set_async_error(c, rc, msg);
        failed = true;
        { std::lock_guard<std::mutex> g(R.mu); R.stop = true; for (auto &kv : R.tasks) kv.second->abort = true; }
        R.cv.notify_all();
        chunks.close(true);
    };

    // ---- frame + pack workers -----------------------------------------------------------------
    // line index, record framing and 2-bit packing run on the device by default (VFIND_FASTQ_DEVICE_FRAMING=0: on these threads)
    const bool device_framing = vf_internal_device_framing(c);
    R.pieces.ctx = c;
    R.pieces.pinned = device_framing;
    std::vector<std::thread> packers;
    // n_threads inflate workers (the reference's n_threads are its alignment workers, next to one reader) and
    // half as many frame + pack workers, whose per-read work is lighter; the environment can override the split
    auto env_u = [](const char *k, uint32_t dflt) { const char *e = getenv(k); const int v = e ? atoi(e) : 0; return v > 0 ? (uint32_t)v : dflt; };
    const uint32_t ninfl = env_u("VFIND_FASTQ_INFLATERS", nthreads);
    R.fq->threads.store(ninfl);
    const uint32_t npack = env_u("VFIND_FASTQ_PACKERS", device_framing ? std::max<uint32_t>(1, (nthreads + 3) / 4) : std::max<uint32_t>(1, (nthreads + 1) / 2));
    std::vector<std::thread> inflaters;
    std::atomic<bool> watch_stop{false};
    std::thread watchdog;
    // An exception on this thread (the sequencer allocates) must not unwind past running workers: a joinable std::thread
    // that is destroyed ends the process.  Whatever way this function is left, the workers are told to stop and joined.
    struct JoinAll {
        std::function<void()> f;
        ~JoinAll() { if (f) f(); }
    } join_all;
    join_all.f = [&] {
        { std::lock_guard<std::mutex> g(R.mu); R.stop = true; for (auto &kv : R.tasks) kv.second->abort = true; }
        R.cv.notify_all();
        for (auto &th : inflaters) if (th.joinable()) th.join();
        watch_stop = true;
        if (watchdog.joinable()) watchdog.join();
        chunks.close(true);
        for (auto &th : packers) if (th.joinable()) th.join();
    };
    // a worker that throws (std::bad_alloc from a buffer, anything else from below) fails the run, it does not end the process
    auto guarded = [&](auto &&body) {
        try { body(); }
        catch (const std::bad_alloc &) { fail_async(VF_ENOMEM, "out of host memory in a FASTQ worker thread"); }
        catch (const std::exception &e) { fail_async(VF_EDEVICE, std::string("internal error in a FASTQ worker thread: ") + e.what()); }
        catch (...) { fail_async(VF_EDEVICE, "internal error in a FASTQ worker thread"); }
    };
    for (uint32_t w = 0; w < npack; ++w) {
        packers.emplace_back([&] { guarded([&] {
            if (vf_internal_test_throw("fastq_worker")) throw std::runtime_error("injected by VFIND_TEST_THROW");
            std::vector<ReadRef> refs;
            std::unique_ptr<Chunk> ch;
            while (chunks.get(ch)) {
                if (failed.load()) continue;
                refs.clear();
                std::string err;
                int rc;
                if (device_framing) {
                    // the device indexes lines, frames records and packs (k_nl_*, k_frame_pack): the host only copies text
                    std::vector<TextSpan> spans;
                    for (auto &sg : ch->segs) spans.push_back(TextSpan{sg.p, sg.n});
                    rc = vf_internal_submit_text(c, spans.data(), spans.size(), ch->final, R.pieces.pinned);
                    ch->segs.clear();
                    if (rc) err = vf_internal_last_error();
                } else {
                // the chunk's byte ranges -> one contiguous buffer (this copy used to sit on the sequencer thread)
                uint8_t *text = chunk_pool.get();
                size_t at = 0;
                if (text) for (auto &sg : ch->segs) { memcpy(text + at, sg.p, sg.n); at += sg.n; }
                ch->segs.clear();                            // pieces return to their pool
                rc = text ? frame_records(text, at, refs, err) : VF_ENOMEM;
                if (!text) err = "out of memory: no buffer for a text chunk";
                if (!rc) {
                    if (g_dbg) fprintf(stderr, "[fastq] packer submits %zu reads at %.3f\n", refs.size(), now_s());
                    rc = vf_internal_submit(c, text, refs.data(), refs.size());
                    if (g_dbg) fprintf(stderr, "[fastq] packer submitted rc=%d at %.3f\n", rc, now_s());
                    if (rc) err = vf_internal_last_error();
                }
                chunk_pool.put(text);
                }
                if (rc) { fail_async(rc, err); break; }
            }
        }); });
    }

    // ---- inflate pool --------------------------------------------------------------------------
    {
        std::lock_guard<std::mutex> g(R.mu);
        R.required = 0;
    }
    for (uint32_t w = 0; w < ninfl; ++w) inflaters.emplace_back([&] { guarded([&] { inflate_worker(R); }); });

    if (g_dbg) watchdog = std::thread([&] {
        while (!watch_stop.load()) {
            for (int i = 0; i < 20 && !watch_stop.load(); ++i) usleep(100000);
            std::lock_guard<std::mutex> g(R.mu);
            fprintf(stderr, "[fastq] watchdog %.1f: head_pos=%llu head_in=%llu required=%lld spec_last=%lld total_buffered=%zu stop=%d tasks:", now_s(),
                    (unsigned long long)R.head_pos, (unsigned long long)R.head_in.load(), (long long)R.required, (long long)R.spec_last, R.total_buffered, (int)R.stop);
            for (auto &kv : R.tasks) fprintf(stderr, " [%llu buf=%zu out=%zu done=%d fail=%d abort=%d in=%llu]", (unsigned long long)kv.first, kv.second->buffered,
                                             kv.second->out.size(), (int)kv.second->done, (int)kv.second->failed, (int)kv.second->abort, (unsigned long long)kv.second->in_pos.load());
            { std::lock_guard<std::mutex> g2(chunks.mu); fprintf(stderr, " chunks_q=%zu closed=%d", chunks.q.size(), (int)chunks.closed); }
            fprintf(stderr, "\n");
        }
    });

    // ---- sequencer (this thread) ----------------------------------------------------------------
    // Bookkeeping only: pieces arrive with their newline counts, so the running line count mod 4 tells how
    // many newlines back from the end of the collected text the last record boundary lies.
    std::unique_ptr<Chunk> cur(new Chunk);
    size_t lines = 0;
    auto emit_full = [&](bool final_flush) -> bool {
        if (cur->size == 0) return true;
        std::unique_ptr<Chunk> next(new Chunk);
        size_t carried_lines = 0;
        if (!final_flush) {                              // (at the very end the last record may lack its newline)
            // cut just after the (k+1)-th newline from the end, k = complete lines past the last whole record
            int need = (int)(lines % 4) + 1;
            int si = (int)cur->segs.size() - 1;
            size_t off = 0;
            bool found = false;
            for (; si >= 0 && !found; --si) {
                const Seg &sg = cur->segs[(size_t)si];
                size_t len = sg.n;
                while (len) {
                    const void *q = memrchr(sg.p, '\n', len);
                    if (!q) break;
                    len = (size_t)((const uint8_t *)q - sg.p);
                    if (--need == 0) { off = len + 1; found = true; break; }
                }
                if (found) break;
            }
            if (!found) return true;                     // less than one whole record so far
            carried_lines = lines % 4;
            // everything after (si, off) moves to the next chunk
            Seg &cut = cur->segs[(size_t)si];
            if (off < cut.n) { Seg tail = cut; tail.p += off; tail.n -= off; next->segs.push_back(tail); next->size += tail.n; }
            for (size_t j = (size_t)si + 1; j < cur->segs.size(); ++j) { next->size += cur->segs[j].n; next->segs.push_back(std::move(cur->segs[j])); }
            cur->segs.resize((size_t)si + 1);
            cur->segs[(size_t)si].n = off;
            cur->size -= next->size;
            if (cur->size == 0) { cur = std::move(next); lines = carried_lines; return true; }
        }
        if (g_dbg) fprintf(stderr, "[fastq] sequencer emits chunk of %zu bytes at %.3f\n", cur->size, now_s());
        cur->final = final_flush;
        const double tq = now_s();
        const bool put_ok = chunks.put(std::move(cur));
        R.fq->queue_full_ns += (uint64_t)((now_s() - tq) * 1e9);
        if (!put_ok) return false;
        cur = std::move(next);
        lines = carried_lines;
        return true;
    };

    bool ok = true;
    while (ok && !failed.load()) {
        std::shared_ptr<Task> t;
        {
            std::unique_lock<std::mutex> lk(R.mu);
            if (R.head_pos >= R.size) break;
            auto it = R.tasks.find(R.head_pos);
            if (it == R.tasks.end()) {
                R.required = R.head_pos;
                R.cv.notify_all();
                R.cv.wait(lk, [&] { return R.stop || R.tasks.count(R.head_pos); });
                if (R.stop) break;
                it = R.tasks.find(R.head_pos);
            }
            t = it->second;
            R.cv.notify_all();                           // the head task may exceed its buffer cap
        }
        // consume the head task's pieces as they arrive
        for (;;) {
            Piece p;
            bool have = false, finished = false;
            {
                std::unique_lock<std::mutex> lk(R.mu);
                if (t->out.empty() && !t->done && !R.stop) {
                    const double tw = now_s();
                    R.cv.wait(lk, [&] { return R.stop || !t->out.empty() || t->done; });
                    R.fq->seq_wait_ns += (uint64_t)((now_s() - tw) * 1e9);
                }
                if (R.stop) { ok = false; break; }
                if (!t->out.empty()) {
                    p = t->out.front();
                    t->out.pop_front();
                    t->buffered -= p.size;
                    R.total_buffered -= p.size;
                    have = true;
                    R.cv.notify_all();
                } else if (t->done) {
                    finished = true;
                }
                // everything that starts inside what the validated stream already consumed is a false start
                const uint64_t consumed = t->in_pos.load(std::memory_order_relaxed);
                R.head_in = consumed;
                for (auto &kv : R.tasks)
                    if (kv.first != t->start && kv.first < consumed && !kv.second->abort) { kv.second->abort = true; }
            }
            if (have) {
                if (cur->size + p.size > kChunkCap) {
                    R.pieces.put(p.data);
                    fail_async(VF_EFORMAT, "malformed FASTQ: a record larger than 16 MiB");
                    ok = false;
                    break;
                }
                Seg sg;
                sg.hold = std::make_shared<PieceHold>(p.data, &R.pieces);
                sg.p = p.data; sg.n = p.size;
                cur->segs.push_back(std::move(sg));
                cur->size += p.size;
                lines += p.nl;
                if (cur->size >= kChunkTarget && !emit_full(false)) { ok = false; break; }
                continue;
            }
            if (finished) {
                if (t->failed) { fail_async(VF_EFORMAT, t->err.empty() ? "gzip error" : t->err); ok = false; }
                break;
            }
        }
        if (!ok) break;
        {
            std::lock_guard<std::mutex> g(R.mu);
            R.head_pos = t->end;
            R.head_in = t->end;
            for (auto it = R.tasks.begin(); it != R.tasks.end();) {
                if (it->first < R.head_pos) {
                    it->second->abort = true;
                    R.total_buffered -= std::min(R.total_buffered, it->second->buffered);
                    it->second->buffered = 0;
                    for (auto &pc : it->second->out) R.pieces.put(pc.data);
                    it->second->out.clear();
                    it = R.tasks.erase(it);
                }
                else ++it;
            }
        }
        R.cv.notify_all();
    }
    if (ok && !failed.load() && cur->size) ok = emit_full(true);
    cur.reset();

    {
        std::lock_guard<std::mutex> g(R.mu);
        R.stop = true;
        for (auto &kv : R.tasks) kv.second->abort = true;
    }
    R.cv.notify_all();
    for (auto &th : inflaters) th.join();
    watch_stop = true;
    if (watchdog.joinable()) watchdog.join();
    chunks.close(false);
    for (auto &th : packers) th.join();
    join_all.f = nullptr;                      // everything is joined
    const double td = now_s();
    const int src = vf_sync(c);
    R.fq->drain_ns += (uint64_t)((now_s() - td) * 1e9);
    return src;
}
