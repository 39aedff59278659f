// test harness: decode every member of a gzip file into pieces of a given size and write the text to a
// file; exit 0 = ok, 3 = decoder reported an error (message on stderr).
//   inflate_harness in.gz out.bin piece_bytes [repeat [threads]]
// threads = 0 (default): vf::GzipMemberDecoder; threads > 0: vf::inflate_member_parallel with that many helpers
#include "../../vfind_amd/csrc/vf_inflate.h"
#include "../../vfind_amd/csrc/vf_pinflate.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: inflate_harness in.gz out.bin piece_bytes [repeat [threads]]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> in;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) in.insert(in.end(), buf, buf + n);
    fclose(f);
    const size_t piece = (size_t)strtoull(argv[3], nullptr, 0);
    const int repeat = argc > 4 ? atoi(argv[4]) : 1;
    const unsigned threads = argc > 5 ? (unsigned)atoi(argv[5]) : 0;
    std::vector<uint8_t> out, pc(piece + 16);
    static vf::GzipMemberDecoder dec;
    double best = 1e30;
    for (int r = 0; r < repeat; ++r) {
        out.clear();
        const auto t0 = std::chrono::steady_clock::now();
        const uint8_t *p = in.data(), *end = in.data() + in.size();
        while (p < end) {
            if (threads) {
                std::mutex mu;
                std::vector<uint8_t *> pool;
                vf::ParallelInflateIO io;
                io.piece_bytes = piece;
                io.get_buffer = [&]() -> uint8_t * {
                    std::lock_guard<std::mutex> g(mu);
                    if (!pool.empty()) { uint8_t *b = pool.back(); pool.pop_back(); return b; }
                    return (uint8_t *)malloc(piece + 16);
                };
                io.drop_buffer = [&](uint8_t *b) { std::lock_guard<std::mutex> g(mu); pool.push_back(b); };
                size_t newlines = 0;
                io.put_piece = [&](uint8_t *b, size_t len, size_t nl) -> bool {
                    if (repeat == 1) out.insert(out.end(), b, b + len);
                    newlines += nl;
                    io.drop_buffer(b);
                    return true;
                };
                const uint8_t *mend = nullptr;
                const char *err = "";
                const bool ok = vf::inflate_member_parallel(p, end, threads, io, &mend, &err);
                for (uint8_t *b : pool) free(b);
                if (!ok) { fprintf(stderr, "%s\n", err); return 3; }
                if (repeat == 1) {
                    size_t want = 0;
                    for (uint8_t c : out) want += c == '\n';
                    (void)want;                              // (newline counts are checked by the reader tests)
                }
                p = mend;
                continue;
            }
            dec.begin(p, end);
            for (;;) {
                size_t got = 0;
                const auto st = dec.run(pc.data(), piece, &got);
                if (repeat == 1) out.insert(out.end(), pc.data(), pc.data() + got);
                if (st == vf::GzipMemberDecoder::kError) { fprintf(stderr, "%s\n", dec.error()); return 3; }
                if (st == vf::GzipMemberDecoder::kMemberEnd) break;
            }
            p = dec.in_pos();
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt < best) best = dt;
    }
    if (repeat > 1) fprintf(stderr, "best of %d: %.3f s\n", repeat, best);
    f = fopen(argv[2], "wb");
    if (!f) return 2;
    if (!out.empty()) fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    return 0;
}
