// vf_synth.cpp -- seeded synthetic read generator (bench / test tooling; see vfind_synth.h).
// Built into benchlib/libvfind_synth.so: not part of the product library, not part of the drop-in boundary.
#include "vfind_synth.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
enum { VF_OK = 0, VF_EINVAL = 1 };
// 2-bit pack of an ACGT-only read (A0 C1 G2 T3, base k at bits [2k, 2k+1] of the row, zero padded); 1 = other byte met
int pack_row(const uint8_t *seq, uint32_t n, uint32_t *row, uint32_t W) {
    for (uint32_t w = 0; w < W; ++w) row[w] = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const uint8_t c = seq[k];
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return 1;
        const uint32_t x = (c >> 1) & 3u;
        row[k >> 4] |= (x ^ (x >> 1)) << (2u * (k & 15u));
    }
    return 0;
}
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {                       // splitmix64
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    uint8_t base() { return (uint8_t)"ACGT"[next() >> 62]; }
};
}  // namespace

struct vf_synth {
    vf_synth_cfg cfg;
    std::string pre, suf;
    std::vector<double> cdf;
};

static bool has_border(const std::string &s) {
    for (size_t k = 1; k < s.size(); ++k)
        if (s.compare(0, s.size() - k, s, k, s.size() - k) == 0) return true;
    return false;
}

extern "C" int vf_synth_adapters(uint64_t seed, uint32_t m, char *p_out, char *s_out) {
    if (!p_out || !s_out || m == 0) return VF_EINVAL;
    Rng r(seed ^ 0xADA97E55ull);
    std::string out[2];
    for (int which = 0; which < 2; ++which) {
        for (int tries = 0; tries < 100000; ++tries) {
            std::string s(m, 'A');
            int gc = 0;
            for (auto &c : s) { c = (char)r.base(); gc += (c == 'G' || c == 'C'); }
            if (m >= 5 && (gc * 10 < (int)m * 4 || gc * 10 > (int)m * 6)) continue;
            if (m >= 4 && has_border(s)) continue;
            if (which == 1 && s == out[0]) continue;
            out[which] = s;
            break;
        }
        if (out[which].empty()) return VF_EINVAL;
    }
    memcpy(p_out, out[0].c_str(), m + 1);
    memcpy(s_out, out[1].c_str(), m + 1);
    return VF_OK;
}

extern "C" int vf_synth_create(const vf_synth_cfg *cfg, vf_synth **out) {
    if (!cfg || !out || !cfg->n_variants || !cfg->read_len) return VF_EINVAL;
    vf_synth *g = new vf_synth;
    g->cfg = *cfg;
    g->pre.assign((const char *)cfg->prefix, cfg->prefix_len);
    g->suf.assign((const char *)cfg->suffix, cfg->suffix_len);
    g->cfg.prefix = (const uint8_t *)g->pre.data();
    g->cfg.suffix = (const uint8_t *)g->suf.data();
    g->cdf.resize(cfg->n_variants);
    double acc = 0;
    for (uint64_t k = 0; k < cfg->n_variants; ++k) { acc += std::pow((double)(k + 1), -cfg->zipf_s); g->cdf[k] = acc; }
    for (auto &v : g->cdf) v /= acc;
    *out = g;
    return VF_OK;
}
extern "C" void vf_synth_destroy(vf_synth *g) { delete g; }

// variant v: length 3k with k uniform in [8, (L-2m-16)/3]; a non_codon_frac share gets +1/+2
static void variant_seq(const vf_synth &g, uint64_t v, std::string &out) {
    Rng r(g.cfg.seed * 0x9E3779B97F4A7C15ull ^ (v + 0x51ED270B1ull));
    const int L = (int)g.cfg.read_len, m2 = (int)(g.pre.size() + g.suf.size());
    int kmax = (L - m2 - 16) / 3;
    if (kmax < 8) kmax = 8;
    int len = 3 * (8 + (int)r.below((uint32_t)(kmax - 8 + 1)));
    if (r.uniform() < g.cfg.non_codon_frac) len += 1 + (int)r.below(2);
    out.resize((size_t)len);
    for (auto &c : out) c = (char)r.base();
}

static void mutate(Rng &r, const std::string &a, double sub, double indel, std::string &out) {
    for (char ch : a) {
        const double u = r.uniform();
        if (u < sub) {                      // substitution by a different base
            char c;
            do { c = (char)r.base(); } while (c == ch);
            out.push_back(c);
        } else if (u < sub + indel * 0.5) { // deletion
        } else if (u < sub + indel) {       // insertion after the base
            out.push_back(ch);
            out.push_back((char)r.base());
        } else out.push_back(ch);
    }
}

static void one_read(const vf_synth &g, uint64_t idx, std::string &buf, std::string &var) {
    Rng r(g.cfg.seed ^ (idx * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull));
    const size_t L = g.cfg.read_len;
    buf.clear();
    if (r.uniform() >= g.cfg.no_adapter_frac) {
        const uint32_t fl = r.below(g.cfg.flank_max + 1);
        for (uint32_t i = 0; i < fl; ++i) buf.push_back((char)r.base());
        mutate(r, g.pre, g.cfg.sub_rate, g.cfg.indel_rate, buf);
        const double u = r.uniform();
        const uint64_t v = (uint64_t)(std::lower_bound(g.cdf.begin(), g.cdf.end(), u) - g.cdf.begin());
        variant_seq(g, std::min<uint64_t>(v, g.cfg.n_variants - 1), var);
        buf += var;
        mutate(r, g.suf, g.cfg.sub_rate, g.cfg.indel_rate, buf);
    }
    while (buf.size() < L) buf.push_back((char)r.base());
    buf.resize(L);
}

extern "C" int vf_synth_generate(const vf_synth *g, uint64_t first, uint64_t n, uint8_t *ascii, uint32_t *rows,
                                 uint32_t W, uint16_t *lens, uint32_t n_threads) {
    if (!g) return VF_EINVAL;
    const uint32_t L = g->cfg.read_len;
    if (rows && L > W * 16u) return VF_EINVAL;
    if (n_threads < 1) n_threads = 1;
    n_threads = (uint32_t)std::min<uint64_t>(n_threads, std::max<uint64_t>(n / 1024, 1));
    std::vector<std::thread> th;
    std::vector<int> rc(n_threads, 0);
    for (uint32_t t = 0; t < n_threads; ++t) {
        th.emplace_back([&, t] {
            std::string buf, var;
            const uint64_t lo = n * t / n_threads, hi = n * (t + 1) / n_threads;
            for (uint64_t i = lo; i < hi; ++i) {
                one_read(*g, first + i, buf, var);
                if (ascii) memcpy(ascii + i * L, buf.data(), L);
                if (rows && pack_row((const uint8_t *)buf.data(), L, rows + i * W, W)) rc[t] = VF_EINVAL;
                if (lens) lens[i] = (uint16_t)L;
            }
        });
    }
    for (auto &x : th) x.join();
    for (int r : rc) if (r) return r;
    return VF_OK;
}
