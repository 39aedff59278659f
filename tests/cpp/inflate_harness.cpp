// test harness: decode every member of a gzip file with vf::GzipMemberDecoder into pieces of a given
// size and write the text to stdout-file; exit 0 = ok, 3 = decoder reported an error (message on stderr)
#include "../../vfind_amd/csrc/vf_inflate.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: inflate_harness in.gz out.bin piece_bytes [repeat]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> in;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) in.insert(in.end(), buf, buf + n);
    fclose(f);
    const size_t piece = (size_t)strtoull(argv[3], nullptr, 0);
    const int repeat = argc > 4 ? atoi(argv[4]) : 1;
    std::vector<uint8_t> out, pc(piece + 16);
    static vf::GzipMemberDecoder dec;
    double best = 1e30;
    for (int r = 0; r < repeat; ++r) {
        out.clear();
        const auto t0 = std::chrono::steady_clock::now();
        const uint8_t *p = in.data(), *end = in.data() + in.size();
        while (p < end) {
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
