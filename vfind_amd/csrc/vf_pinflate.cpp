// vf_pinflate.cpp -- one gzip member decoded by several threads (see vf_pinflate.h)
#include "vf_pinflate.h"

#include "vf_inflate.h"

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace vf {

namespace {

size_t cell_bytes() {                                    // compressed bytes per cell (tests shrink it)
    static const size_t v = [] { const char *e = getenv("VFIND_PINFLATE_CELL"); const long long x = e ? atoll(e) : 0; return x >= 4096 ? (size_t)x : (size_t)(1u << 20); }();
    return v;
}

// growable symbol buffer without value-initialisation
struct SymBuf {
    uint16_t *p = nullptr;
    size_t n = 0, cap = 0;
    SymBuf() = default;
    SymBuf(const SymBuf &) = delete;
    SymBuf &operator=(const SymBuf &) = delete;
    SymBuf(SymBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    SymBuf &operator=(SymBuf &&o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; } return *this; }
    ~SymBuf() { free(p); }
    bool room(size_t want) {
        if (cap - n >= want) return true;
        size_t nc = cap ? cap * 2 : (size_t)8 << 20;
        while (nc - n < want) nc *= 2;
        uint16_t *q = (uint16_t *)realloc(p, nc * sizeof(uint16_t));
        if (!q) return false;
        p = q; cap = nc;
        return true;
    }
};

struct CellResult {
    bool found = false, ended = false;
    uint64_t start_bit = 0, end_bit = 0;
    SymBuf sym;
};

inline uint64_t peek_bits(const uint8_t *base, const uint8_t *end, uint64_t bit, int n) {   // n <= 32
    const uint8_t *b = base + (bit >> 3);
    uint64_t v = 0;
    if (b >= end) return 0;
    const size_t avail = (size_t)(end - b);
    if (avail >= 8) memcpy(&v, b, 8); else memcpy(&v, b, avail);
    return (v >> (bit & 7u)) & ((1ull << n) - 1ull);
}

// structural test of a dynamic-Huffman block header at bit position p: not the last block, symbol counts in
// range, and the code-length code complete (Kraft sum exactly 1) - about one random position in 10^3 passes
inline bool plausible_header(const uint8_t *base, const uint8_t *end, uint64_t p) {
    if ((uint64_t)(end - base) * 8u < p + 17 + 19 * 3 + 64) return false;      // a real block this close to the end is left to the coordinator
    const uint64_t h = peek_bits(base, end, p, 17);
    if ((h & 7u) != 4u) return false;                    // BFINAL = 0, BTYPE = 2
    if (((h >> 3) & 31u) > 29u || ((h >> 8) & 31u) > 29u) return false;
    const int hclen = (int)((h >> 13) & 15u) + 4;
    int sum = 0, nz = 0, mx = 0;
    uint64_t q = p + 17;
    for (int i = 0; i < hclen; i += 8) {
        const int take = std::min(8, hclen - i);
        uint64_t v = peek_bits(base, end, q, 3 * take);
        for (int j = 0; j < take; ++j) { const int l = (int)(v & 7u); v >>= 3; if (l) { sum += 128 >> l; ++nz; if (l > mx) mx = l; } }
        q += (uint64_t)(3 * take);
    }
    return sum == 128 || (nz == 1 && mx == 1);
}

// first plausible block start in [lo_bit, hi_bit) whose first block decodes cleanly; decoded on to the first
// block boundary at or beyond hi_bit (or the end of the stream), window unknown
CellResult run_cell(const uint8_t *base, const uint8_t *end, uint64_t lo_bit, uint64_t hi_bit, DeflateCore<uint16_t> &core) {
    using Core = DeflateCore<uint16_t>;
    CellResult res;
    auto decode = [&](Core::Status &st, size_t limit) -> bool {      // false = out of memory / runaway
        for (;;) {
            if (!res.sym.room(1u << 16)) return false;
            size_t got = 0;
            st = core.run(res.sym.p + res.sym.n, res.sym.cap - res.sym.n, &got);
            res.sym.n += got;
            if (st != Core::kPieceFull) return true;
            if (res.sym.n > limit) return false;
        }
    };
    for (uint64_t p = lo_bit; p < hi_bit; ++p) {
        if (!plausible_header(base, end, p)) continue;
        core.begin(base, end, p);
        core.fill_markers();
        core.set_stop_bit(p + 1);                         // the end of the first block
        res.sym.n = 0;
        Core::Status st;
        if (!decode(st, (size_t)64 << 20)) continue;
        if ((st != Core::kBoundary && st != Core::kStreamEnd) || res.sym.n == 0) continue;
        if (st == Core::kBoundary) {
            core.set_stop_bit(hi_bit);
            if (!decode(st, (size_t)1 << 31)) continue;
            if (st != Core::kBoundary && st != Core::kStreamEnd) continue;      // a later block is broken: this was no real start
        }
        res.found = true;
        res.start_bit = p;
        res.end_bit = core.bit_pos();
        res.ended = st == Core::kStreamEnd;
        return res;
    }
    res.sym = SymBuf();
    return res;
}

struct Pool {
    std::vector<std::thread> th;
    std::deque<std::function<void()>> q;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    explicit Pool(unsigned n) {
        for (unsigned i = 0; i < n; ++i) th.emplace_back([this] {
            for (;;) {
                std::function<void()> f;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || !q.empty(); });
                    if (q.empty()) return;
                    f = std::move(q.front());
                    q.pop_front();
                }
                f();
            }
        });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
    template <class F> auto run(F f) -> std::future<decltype(f())> {
        auto task = std::make_shared<std::packaged_task<decltype(f())()>>(std::move(f));
        auto fut = task->get_future();
        { std::lock_guard<std::mutex> g(mu); q.emplace_back([task] { (*task)(); }); }
        cv.notify_one();
        return fut;
    }
};

// symbols -> bytes: literals pass, markers are looked up in the window W that preceded the cell.  Past the
// first 32 KiB of a cell nearly everything is a literal, so 16 symbols at a time are narrowed with one
// compare + pack when none of them is a marker.
inline void resolve_symbols(const uint16_t *s, size_t n, const uint8_t *W, uint8_t *out) {
    size_t i = 0;
#if defined(__SSE2__)
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(s + i)), b = _mm_loadu_si128((const __m128i *)(s + i + 8));
        // unsigned v > 255  <=>  (v >> 8) != 0
        const __m128i hi = _mm_or_si128(_mm_srli_epi16(a, 8), _mm_srli_epi16(b, 8));
        if (_mm_movemask_epi8(_mm_cmpeq_epi8(hi, _mm_setzero_si128())) == 0xFFFF) {
            _mm_storeu_si128((__m128i *)(out + i), _mm_packus_epi16(a, b));
        } else {
            for (size_t j = i; j < i + 16; ++j) { const uint16_t v = s[j]; out[j] = v < kMarkerBase ? (uint8_t)v : W[v - kMarkerBase]; }
        }
    }
#endif
    for (; i < n; ++i) { const uint16_t v = s[i]; out[i] = v < kMarkerBase ? (uint8_t)v : W[v - kMarkerBase]; }
}

struct OutPiece { uint8_t *buf; size_t n, nl; };
struct Resolved { std::vector<OutPiece> pieces; uint32_t crc = 0; uint64_t len = 0; bool oom = false; };

}  // namespace

bool inflate_member_parallel(const uint8_t *in, const uint8_t *in_end, unsigned threads, const ParallelInflateIO &io,
                             const uint8_t **member_end, const char **err) {
    using Core8 = DeflateCore<uint8_t>;
    *err = "";
    const size_t hdr = GzipMemberDecoder::parse_header(in, in_end, err);
    if (!hdr) return false;
    const uint8_t *base = in;
    const uint64_t size = (uint64_t)(in_end - in);
    const uint64_t cell = cell_bytes();
    const uint64_t n_cells = (size + cell - 1) / cell;
    const unsigned helpers = std::max(1u, threads);
    const uint64_t ahead = 2ull * helpers;

    uint64_t cur = (uint64_t)hdr * 8u;                     // bit position all emitted / scheduled text ends on
    std::vector<uint8_t> window;                           // the (up to) 32 KiB of text in front of cur
    uint32_t crc = 0;
    uint64_t total = 0;
    bool aborted = false, out_of_memory = false;
    static const char *const kOom = "out of memory: no buffer for inflated text";

    std::unique_ptr<Core8> core8(new Core8);
    std::vector<std::unique_ptr<DeflateCore<uint16_t>>> cores16;
    std::mutex cores_mu;
    auto borrow16 = [&]() -> std::unique_ptr<DeflateCore<uint16_t>> {
        std::lock_guard<std::mutex> g(cores_mu);
        if (cores16.empty()) return std::unique_ptr<DeflateCore<uint16_t>>(new DeflateCore<uint16_t>);
        auto c = std::move(cores16.back());
        cores16.pop_back();
        return c;
    };
    auto give16 = [&](std::unique_ptr<DeflateCore<uint16_t>> c) { std::lock_guard<std::mutex> g(cores_mu); cores16.push_back(std::move(c)); };

    std::map<uint64_t, std::future<CellResult>> cells;
    std::deque<std::future<Resolved>> pending;
    Pool pool(helpers);                                    // declared last: joins before the state above goes away

    auto emit_front = [&]() -> bool {
        Resolved r = pending.front().get();
        pending.pop_front();
        crc = crc32_join(crc, r.crc, r.len);
        total += r.len;
        if (r.oom) { aborted = true; out_of_memory = true; }
        for (auto &pc : r.pieces) {
            if (aborted) { io.drop_buffer(pc.buf); continue; }
            if (!io.put_piece(pc.buf, pc.n, pc.nl)) aborted = true;
        }
        return !aborted;
    };
    auto emit_pending = [&](bool all) -> bool {
        while (!pending.empty() &&
               (all || pending.size() > ahead || pending.front().wait_for(std::chrono::seconds(0)) == std::future_status::ready))
            if (!emit_front()) { while (!pending.empty()) (void)emit_front(); return false; }
        return true;
    };

    // sequential decoding from cur (window known) to the first block boundary at or beyond stop_bit
    auto exact_until = [&](uint64_t stop_bit, Core8::Status &st) -> bool {
        st = Core8::kBoundary;
        if (cur >= stop_bit) return true;
        if (!emit_pending(true)) { *err = out_of_memory ? kOom : "aborted"; return false; }
        core8->begin(base, in_end, cur);
        core8->set_history(window.data(), window.size());
        core8->set_stop_bit(stop_bit);
        for (;;) {
            uint8_t *buf = io.get_buffer();
            if (!buf) { *err = "out of memory: no buffer for inflated text"; return false; }
            size_t got = 0;
            st = core8->run(buf, io.piece_bytes, &got);
            if (got) {
                crc = crc32_update(crc, buf, got);
                total += got;
                if (!io.put_piece(buf, got, count_byte(buf, got, '\n'))) { aborted = true; *err = "aborted"; return false; }
            } else {
                io.drop_buffer(buf);
            }
            if (st == Core8::kError) { *err = core8->error(); return false; }
            if (st != Core8::kPieceFull) break;
        }
        cur = core8->bit_pos();
        size_t hn = 0;
        const uint8_t *h = core8->history(&hn);
        window.assign(h, h + hn);
        if (io.progress) io.progress(cur / 8);
        return true;
    };

    uint64_t next_cell = cur / 8 / cell + 1;               // first cell that starts beyond cur
    uint64_t submitted_to = next_cell;
    auto top_up = [&] {
        if (submitted_to < next_cell) submitted_to = next_cell;
        while (submitted_to < n_cells && submitted_to < next_cell + ahead) {
            const uint64_t k = submitted_to++;
            const uint64_t lo = k * cell * 8u, hi = std::min<uint64_t>((k + 1) * cell, size) * 8u;
            cells[k] = pool.run([&, lo, hi]() -> CellResult {
                auto c = borrow16();
                CellResult r = run_cell(base, in_end, lo, hi, *c);
                give16(std::move(c));
                return r;
            });
        }
    };

    bool ended = false, ok = true;
    while (!ended) {
        top_up();
        Core8::Status st;
        if (next_cell >= n_cells) {                        // the tail of the stream: no cell starts beyond it
            if (!exact_until(~0ull, st)) { ok = false; break; }
            if (st != Core8::kStreamEnd) { *err = Core8::truncated(); ok = false; break; }
            ended = true;
            break;
        }
        if (!exact_until(next_cell * cell * 8u, st)) { ok = false; break; }
        if (st == Core8::kStreamEnd) { ended = true; break; }
        while (next_cell < n_cells && (next_cell + 1) * cell * 8u <= cur) { cells.erase(next_cell); ++next_cell; }      // cells the text already covers
        if (next_cell >= n_cells) continue;
        top_up();
        CellResult res = cells[next_cell].get();
        cells.erase(next_cell);
        ++next_cell;
        if (!res.found || res.start_bit < cur || window.size() < kDeflateWindow) continue;
        if (res.start_bit > cur) {                         // e.g. a stored or fixed block came first: walk up to the claimed start
            if (!exact_until(res.start_bit, st)) { ok = false; break; }
            if (st == Core8::kStreamEnd) { ended = true; break; }
            if (cur != res.start_bit || window.size() < kDeflateWindow) continue;      // the claim was no block boundary
        }
        // accepted: this cell's symbols continue the text exactly
        std::shared_ptr<std::vector<uint8_t>> w(new std::vector<uint8_t>(window));
        const size_t n = res.sym.n;
        {   // the window behind this cell's text (needs only its last 32 KiB)
            std::vector<uint8_t> nw(kDeflateWindow);
            if (n >= kDeflateWindow) {
                for (size_t i = 0; i < kDeflateWindow; ++i) { const uint16_t v = res.sym.p[n - kDeflateWindow + i]; nw[i] = v < kMarkerBase ? (uint8_t)v : (*w)[v - kMarkerBase]; }
            } else {
                memcpy(nw.data(), w->data() + n, kDeflateWindow - n);
                for (size_t i = 0; i < n; ++i) { const uint16_t v = res.sym.p[i]; nw[kDeflateWindow - n + i] = v < kMarkerBase ? (uint8_t)v : (*w)[v - kMarkerBase]; }
            }
            window.swap(nw);
        }
        cur = res.end_bit;
        ended = res.ended;
        if (io.progress) io.progress(cur / 8);
        std::shared_ptr<SymBuf> sym(new SymBuf(std::move(res.sym)));
        pending.push_back(pool.run([&io, sym, w]() -> Resolved {
            Resolved r;
            const uint8_t *W = w->data();
            size_t at = 0;
            while (at < sym->n) {
                uint8_t *buf = io.get_buffer();
                if (!buf) { r.oom = true; break; }           // reported by the coordinator (emit_front)
                const size_t take = std::min(io.piece_bytes, sym->n - at);
                const uint16_t *s = sym->p + at;
                resolve_symbols(s, take, W, buf);
                r.crc = crc32_update(r.crc, buf, take);
                r.pieces.push_back(OutPiece{buf, take, count_byte(buf, take, '\n')});
                at += take;
            }
            r.len = sym->n;
            return r;
        }));
        if (!emit_pending(false)) { *err = out_of_memory ? kOom : "aborted"; ok = false; break; }
    }
    if (ok && !emit_pending(true)) { *err = out_of_memory ? kOom : "aborted"; ok = false; }
    if (!ok) {
        while (!pending.empty()) { aborted = true; (void)emit_front(); }
        return false;
    }
    const uint8_t *t = base + (cur + 7) / 8;
    if (in_end - t < 8) { *err = Core8::truncated(); return false; }
    const uint32_t want_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    const uint32_t want_size = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
    if (want_crc != crc) { *err = "gzip error: incorrect data check"; return false; }
    if (want_size != (uint32_t)total) { *err = "gzip error: incorrect length check"; return false; }
    *member_end = t + 8;
    return true;
}

}  // namespace vf
