// vf_inflate.cpp -- deflate / gzip decoding (see vf_inflate.h)
#include "vf_inflate.h"

#include <cstring>

namespace vf {

namespace {

// decode-table entry: [7:0] bits to consume, [11:8] extra bits (length / distance symbols) or subtable
// index bits (links), [15:12] flags, [31:16] literal byte / base value / subtable offset
constexpr uint32_t F_LIT = 0x1000, F_EOB = 0x2000, F_SUB = 0x4000, F_BAD = 0x8000;

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint32_t reverse_bits(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}
inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

}  // namespace

template <typename T>
void DeflateCore<T>::begin(const uint8_t *base, const uint8_t *in_end, uint64_t bit_pos) {
    base_ = base; in_end_ = in_end;
    in_ = base + (bit_pos >> 3);
    if (in_ > in_end_) in_ = in_end_;
    bitbuf_ = 0; bitcnt_ = 0;
    const int skip = (int)(bit_pos & 7u);
    if (skip && in_ < in_end_) { bitbuf_ = (uint64_t)*in_++ >> skip; bitcnt_ = 8 - skip; }
    stop_bit_ = ~0ull;
    state_ = kBlockHeader;
    last_block_ = false;
    stored_left_ = 0;
    hist_len_ = 0;
    err_ = "";
}

template <typename T>
void DeflateCore<T>::set_history(const T *text, size_t n) {
    if (n > kDeflateWindow) { text += n - kDeflateWindow; n = kDeflateWindow; }
    if (n) memmove(hist_, text, n * sizeof(T));
    hist_len_ = n;
}

template <typename T>
void DeflateCore<T>::fill_markers() {
    for (size_t i = 0; i < kDeflateWindow; ++i) hist_[i] = (T)(kMarkerBase + i);
    hist_len_ = kDeflateWindow;
}

template <typename T>
bool DeflateCore<T>::need_bits(int n) {
    while (bitcnt_ < n) {
        if (in_ >= in_end_) return false;
        bitbuf_ |= (uint64_t)*in_++ << bitcnt_;
        bitcnt_ += 8;
    }
    return true;
}

// canonical Huffman code -> decode table with per-prefix subtables
template <typename T>
bool DeflateCore<T>::build(const uint8_t *lens, int nsyms, bool is_dist, uint32_t *table, int main_bits, int cap) {
    int count[16] = {0};
    for (int s = 0; s < nsyms; ++s) count[lens[s]]++;
    const int main_size = 1 << main_bits;
    int left = 1, max_len = 0;
    if (count[0] != nsyms) {                               // validate before any table is touched (cheap rejection)
        for (int len = 1; len <= 15; ++len) {
            left <<= 1;
            left -= count[len];
            if (left < 0) return false;                    // over-subscribed
            if (count[len]) max_len = len;
        }
        if (left > 0 && max_len != 1) return false;        // incomplete (zlib accepts only a lone 1-bit code)
    }
    for (int i = 0; i < main_size; ++i) table[i] = F_BAD | 1u;
    if (count[0] == nsyms) return true;                    // no codes at all: any use is an error
    uint32_t first[16];
    uint32_t code = 0;
    count[0] = 0;
    for (int len = 1; len <= 15; ++len) { code = (code + (uint32_t)count[len - 1]) << 1; first[len] = code; }

    auto entry = [&](int sym, int nbits) -> uint32_t {
        if (is_dist) {
            if (sym >= 30) return F_BAD | (uint32_t)nbits;
            return ((uint32_t)kDistBase[sym] << 16) | ((uint32_t)kDistExtra[sym] << 8) | (uint32_t)nbits;
        }
        if (sym < 256) return ((uint32_t)sym << 16) | F_LIT | (uint32_t)nbits;
        if (sym == 256) return F_EOB | (uint32_t)nbits;
        if (sym >= 286) return F_BAD | (uint32_t)nbits;
        return ((uint32_t)kLenBase[sym - 257] << 16) | ((uint32_t)kLenExtra[sym - 257] << 8) | (uint32_t)nbits;
    };

    // pass 1: subtable width per primary prefix
    uint8_t sub_bits[1 << kLitBits];
    bool any_long = max_len > main_bits;
    if (any_long) {
        memset(sub_bits, 0, (size_t)main_size);
        uint32_t next[16];
        memcpy(next, first, sizeof(next));
        for (int s = 0; s < nsyms; ++s) {
            const int len = lens[s];
            if (!len) continue;
            const uint32_t rev = reverse_bits(next[len]++, len);
            if (len > main_bits) {
                const uint32_t prefix = rev & (uint32_t)(main_size - 1);
                if (len - main_bits > sub_bits[prefix]) sub_bits[prefix] = (uint8_t)(len - main_bits);
            }
        }
        int alloc = main_size;
        for (int p = 0; p < main_size; ++p) {
            if (!sub_bits[p]) continue;
            const int size = 1 << sub_bits[p];
            if (alloc + size > cap) return false;
            table[p] = ((uint32_t)alloc << 16) | F_SUB | ((uint32_t)sub_bits[p] << 8) | (uint32_t)main_bits;
            for (int i = 0; i < size; ++i) table[alloc + i] = F_BAD | 1u;
            alloc += size;
        }
    }
    // pass 2: fill
    uint32_t next[16];
    memcpy(next, first, sizeof(next));
    for (int s = 0; s < nsyms; ++s) {
        const int len = lens[s];
        if (!len) continue;
        const uint32_t rev = reverse_bits(next[len]++, len);
        if (len <= main_bits) {
            const uint32_t e = entry(s, len);
            for (uint32_t i = rev; i < (uint32_t)main_size; i += 1u << len) table[i] = e;
        } else {
            const uint32_t prefix = rev & (uint32_t)(main_size - 1);
            const uint32_t link = table[prefix];
            const uint32_t off = link >> 16, sb = (link >> 8) & 15u;
            const uint32_t e = entry(s, len - main_bits);
            for (uint32_t i = rev >> main_bits; i < (1u << sb); i += 1u << (len - main_bits)) table[off + i] = e;
        }
    }
    return true;
}

// RFC 1951 block header: leaves state_ at kStored or kHuffman
template <typename T>
bool DeflateCore<T>::read_block_header() {
    if (!need_bits(3)) { err_ = truncated(); return false; }
    last_block_ = (bitbuf_ & 1u) != 0;
    const uint32_t type = (uint32_t)(bitbuf_ >> 1) & 3u;
    bitbuf_ >>= 3; bitcnt_ -= 3;
    if (type == 0) {
        // stored: to the byte boundary, hand whole bytes back to the input
        const int drop = bitcnt_ & 7;
        bitbuf_ >>= drop; bitcnt_ -= drop;
        in_ -= bitcnt_ >> 3;
        bitbuf_ = 0; bitcnt_ = 0;
        if (in_end_ - in_ < 4) { err_ = truncated(); return false; }
        const uint32_t len = (uint32_t)in_[0] | ((uint32_t)in_[1] << 8), nlen = (uint32_t)in_[2] | ((uint32_t)in_[3] << 8);
        if ((len ^ 0xFFFFu) != nlen) { err_ = "gzip error: invalid stored block lengths"; return false; }
        in_ += 4;
        stored_left_ = len;
        state_ = kStored;
        return true;
    }
    if (type == 3) { err_ = "gzip error: invalid block type"; return false; }
    uint8_t lens[288 + 32];
    int hlit, hdist;
    if (type == 1) {
        hlit = 288; hdist = 32;
        for (int i = 0; i < 144; ++i) lens[i] = 8;
        for (int i = 144; i < 256; ++i) lens[i] = 9;
        for (int i = 256; i < 280; ++i) lens[i] = 7;
        for (int i = 280; i < 288; ++i) lens[i] = 8;
        for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
    } else {
        if (!need_bits(14)) { err_ = truncated(); return false; }
        hlit = (int)(bitbuf_ & 31u) + 257;
        hdist = (int)((bitbuf_ >> 5) & 31u) + 1;
        const int hclen = (int)((bitbuf_ >> 10) & 15u) + 4;
        bitbuf_ >>= 14; bitcnt_ -= 14;
        if (hlit > 286 || hdist > 30) { err_ = "gzip error: too many length or distance symbols"; return false; }
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; ++i) {
            if (!need_bits(3)) { err_ = truncated(); return false; }
            cl[kClOrder[i]] = (uint8_t)(bitbuf_ & 7u);
            bitbuf_ >>= 3; bitcnt_ -= 3;
        }
        // the code-length code: 7-bit table, no subtables needed (lengths <= 7)
        uint32_t cltab[128];
        {
            int count[8] = {0};
            for (int s = 0; s < 19; ++s) count[cl[s]]++;
            int left = 1, mx = 0;
            for (int len = 1; len <= 7; ++len) { left <<= 1; left -= count[len]; if (left < 0) { err_ = "gzip error: invalid code lengths set"; return false; } if (count[len]) mx = len; }
            if (mx == 0 || (left > 0 && mx != 1)) { err_ = "gzip error: invalid code lengths set"; return false; }
            uint32_t first[8], code = 0;
            count[0] = 0;
            for (int len = 1; len <= 7; ++len) { code = (code + (uint32_t)count[len - 1]) << 1; first[len] = code; }
            for (int i = 0; i < 128; ++i) cltab[i] = 0xFFFFFFFFu;
            for (int s = 0; s < 19; ++s) {
                const int len = cl[s];
                if (!len) continue;
                const uint32_t rev = reverse_bits(first[len]++, len);
                for (uint32_t i = rev; i < 128u; i += 1u << len) cltab[i] = ((uint32_t)s << 8) | (uint32_t)len;
            }
        }
        int n = 0;
        const int total = hlit + hdist;
        while (n < total) {
            (void)need_bits(7 + 7);                         // code + the longest repeat field; a short tail is checked below
            const uint32_t e = cltab[bitbuf_ & 127u];
            if (e == 0xFFFFFFFFu) { err_ = "gzip error: invalid code lengths set"; return false; }
            const int nb = (int)(e & 0xFFu), sym = (int)(e >> 8);
            int extra = 0, rep_base = 0;
            if (sym == 16) { extra = 2; rep_base = 3; } else if (sym == 17) { extra = 3; rep_base = 3; } else if (sym == 18) { extra = 7; rep_base = 11; }
            if (bitcnt_ < nb + extra) { err_ = truncated(); return false; }
            bitbuf_ >>= nb; bitcnt_ -= nb;
            if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
            const int rep = rep_base + (int)(bitbuf_ & ((1u << extra) - 1u));
            bitbuf_ >>= extra; bitcnt_ -= extra;
            uint8_t v = 0;
            if (sym == 16) {
                if (n == 0) { err_ = "gzip error: invalid bit length repeat"; return false; }
                v = lens[n - 1];
            }
            if (n + rep > total) { err_ = "gzip error: invalid bit length repeat"; return false; }
            for (int i = 0; i < rep; ++i) lens[n++] = v;
        }
        if (lens[256] == 0) { err_ = "gzip error: invalid code -- missing end-of-block"; return false; }
        // distance lengths follow the literal/length lengths: move them to a fixed place
        uint8_t dl[32] = {0};
        memcpy(dl, lens + hlit, (size_t)hdist);
        memset(lens + hlit, 0, (size_t)(288 - hlit));
        memcpy(lens + 288, dl, 32);
    }
    if (!build(lens, type == 1 ? 288 : hlit, false, lit_, kLitBits, kLitTableSize)) { err_ = "gzip error: invalid literal/lengths set"; return false; }
    if (!build(lens + 288, type == 1 ? 32 : hdist, true, dist_, kDistBits, kDistTableSize)) { err_ = "gzip error: invalid distances set"; return false; }
    state_ = kHuffman;
    return true;
}

// keep the last 32 KiB of text for matches that reach back from the next piece
template <typename T>
void DeflateCore<T>::keep_history(const T *out, size_t pos) {
    if (pos >= kDeflateWindow) {
        memcpy(hist_, out + pos - kDeflateWindow, kDeflateWindow * sizeof(T));
        hist_len_ = kDeflateWindow;
    } else if (pos) {
        size_t keep = hist_len_;
        if (keep > kDeflateWindow - pos) keep = kDeflateWindow - pos;
        memmove(hist_, hist_ + hist_len_ - keep, keep * sizeof(T));
        memcpy(hist_ + keep, out, pos * sizeof(T));
        hist_len_ = keep + pos;
    }
}

template <typename T>
typename DeflateCore<T>::Status DeflateCore<T>::run(T *out, size_t cap, size_t *produced) {
    size_t pos = 0;
    Status result = kError;
    constexpr uint64_t lit_mask = (1u << kLitBits) - 1, dist_mask = (1u << kDistBits) - 1;
    constexpr uint32_t kWordElems = 8 / sizeof(T);

    // copy a match of `len` elements from `dist` back; the source may start in the history of earlier pieces
    auto copy_match = [&](uint32_t len, uint32_t dist) -> bool {
        T *dst = out + pos;
        if (dist > pos) {
            const size_t back = dist - pos;                 // elements before this piece
            if (back > hist_len_) return false;
            size_t i = 0;
            for (; i < len && i < back; ++i) dst[i] = hist_[hist_len_ - back + i];
            const T *src = dst - dist;                      // valid from index `back` on
            for (; i < len; ++i) dst[i] = src[i];
        } else if (dist >= kWordElems) {
            const T *src = dst - dist;
            uint32_t i = 0;
            do { memcpy(dst + i, src + i, 8); i += kWordElems; } while (i < len);      // may overshoot by up to 7 bytes (slack below)
        } else if (dist == 1) {
            const T v = dst[-1];
            if (sizeof(T) == 1) memset(dst, (int)v, len);
            else for (uint32_t i = 0; i < len; ++i) dst[i] = v;
        } else {
            const T *src = dst - dist;
            for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];
        }
        pos += len;
        return true;
    };

    for (;;) {
        if (state_ == kBlockHeader) {
            if (bit_pos() >= stop_bit_) { result = kBoundary; break; }
            if (!read_block_header()) break;
        } else if (state_ == kDone) {
            result = kStreamEnd;
            break;
        } else if (state_ == kStored) {
            size_t n = stored_left_;
            if (n > cap - pos) n = cap - pos;
            if (n > (size_t)(in_end_ - in_)) { err_ = truncated(); break; }
            if (sizeof(T) == 1) memcpy(out + pos, in_, n);
            else for (size_t i = 0; i < n; ++i) out[pos + i] = (T)in_[i];
            in_ += n; pos += n; stored_left_ -= (uint32_t)n;
            if (stored_left_ == 0) state_ = last_block_ ? kDone : kBlockHeader;
            else { result = kPieceFull; break; }
        } else {
            // ---- Huffman block: fast loop while input and output have slack ----
            bool block_done = false, bad = false;
            while (in_end_ - in_ >= 16 && cap - pos >= 274) {
                bitbuf_ |= load64(in_) << bitcnt_;
                in_ += (63 - bitcnt_) >> 3;
                bitcnt_ |= 56;
                uint32_t e = lit_[bitbuf_ & lit_mask];
                if (e & F_SUB) {
                    bitbuf_ >>= kLitBits; bitcnt_ -= kLitBits;
                    e = lit_[(e >> 16) + (uint32_t)(bitbuf_ & ((1u << ((e >> 8) & 15u)) - 1u))];
                }
                bitbuf_ >>= (e & 0xFFu); bitcnt_ -= (int)(e & 0xFFu);
                if (e & F_LIT) {
                    out[pos++] = (T)(e >> 16);
                    // up to two more literals from the bits already in hand (>= 41 left, direct entries take <= 11)
                    e = lit_[bitbuf_ & lit_mask];
                    if ((e & (F_LIT | F_SUB)) == F_LIT) {
                        bitbuf_ >>= (e & 0xFFu); bitcnt_ -= (int)(e & 0xFFu);
                        out[pos++] = (T)(e >> 16);
                        e = lit_[bitbuf_ & lit_mask];
                        if ((e & (F_LIT | F_SUB)) == F_LIT) {
                            bitbuf_ >>= (e & 0xFFu); bitcnt_ -= (int)(e & 0xFFu);
                            out[pos++] = (T)(e >> 16);
                        }
                    }
                    continue;
                }
                if (e & (F_EOB | F_BAD)) {
                    if (e & F_BAD) { err_ = "gzip error: invalid literal/length code"; bad = true; }
                    block_done = true;
                    break;
                }
                uint32_t xb = (e >> 8) & 15u;
                const uint32_t len = (e >> 16) + (uint32_t)(bitbuf_ & ((1u << xb) - 1u));
                bitbuf_ >>= xb; bitcnt_ -= (int)xb;
                uint32_t d = dist_[bitbuf_ & dist_mask];
                if (d & F_SUB) {
                    bitbuf_ >>= kDistBits; bitcnt_ -= kDistBits;
                    d = dist_[(d >> 16) + (uint32_t)(bitbuf_ & ((1u << ((d >> 8) & 15u)) - 1u))];
                }
                if (d & F_BAD) { err_ = "gzip error: invalid distance code"; bad = true; block_done = true; break; }
                bitbuf_ >>= (d & 0xFFu); bitcnt_ -= (int)(d & 0xFFu);
                xb = (d >> 8) & 15u;
                const uint32_t dist = (d >> 16) + (uint32_t)(bitbuf_ & ((1u << xb) - 1u));
                bitbuf_ >>= xb; bitcnt_ -= (int)xb;
                if (!copy_match(len, dist)) { err_ = "gzip error: invalid distance too far back"; bad = true; block_done = true; break; }
            }
            if (bad) break;
            if (!block_done) {
                // ---- careful loop: near the end of the input or of the piece ----
                for (;;) {
                    if (cap - pos < 274) { result = kPieceFull; goto out; }
                    while (bitcnt_ <= 56 && in_ < in_end_) { bitbuf_ |= (uint64_t)*in_++ << bitcnt_; bitcnt_ += 8; }
                    uint32_t e = lit_[bitbuf_ & lit_mask];
                    int used = 0;
                    uint64_t bb = bitbuf_;
                    if (e & F_SUB) {
                        bb >>= kLitBits; used += kLitBits;
                        e = lit_[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 15u)) - 1u))];
                    }
                    used += (int)(e & 0xFFu); bb >>= (e & 0xFFu);
                    if (e & F_BAD) { err_ = used > bitcnt_ ? truncated() : "gzip error: invalid literal/length code"; goto out; }
                    if (used > bitcnt_) { err_ = truncated(); goto out; }
                    if (e & F_LIT) { bitbuf_ = bb; bitcnt_ -= used; out[pos++] = (T)(e >> 16); continue; }
                    if (e & F_EOB) { bitbuf_ = bb; bitcnt_ -= used; block_done = true; break; }
                    uint32_t xb = (e >> 8) & 15u;
                    const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << xb) - 1u));
                    bb >>= xb; used += (int)xb;
                    uint32_t d = dist_[bb & dist_mask];
                    if (d & F_SUB) {
                        bb >>= kDistBits; used += kDistBits;
                        d = dist_[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 15u)) - 1u))];
                    }
                    if (d & F_BAD) { err_ = used > bitcnt_ ? truncated() : "gzip error: invalid distance code"; goto out; }
                    bb >>= (d & 0xFFu); used += (int)(d & 0xFFu);
                    xb = (d >> 8) & 15u;
                    const uint32_t dist = (d >> 16) + (uint32_t)(bb & ((1u << xb) - 1u));
                    bb >>= xb; used += (int)xb;
                    if (used > bitcnt_) { err_ = truncated(); goto out; }
                    bitbuf_ = bb; bitcnt_ -= used;
                    if (!copy_match(len, dist)) { err_ = "gzip error: invalid distance too far back"; goto out; }
                }
            }
            state_ = last_block_ ? kDone : kBlockHeader;
        }
    }
out:
    keep_history(out, pos);
    *produced = pos;
    return result;
}

template class DeflateCore<uint8_t>;
template class DeflateCore<uint16_t>;

// ---- gzip member framing ---------------------------------------------------------------------------
size_t GzipMemberDecoder::parse_header(const uint8_t *p0, const uint8_t *end, const char **err) {
    const uint8_t *p = p0;
    const char *trunc = DeflateCore<uint8_t>::truncated();
    if (end - p < 10) { *err = trunc; return 0; }
    if (p[0] != 0x1f || p[1] != 0x8b) { *err = "gzip error: invalid gzip header"; return 0; }
    if (p[2] != 8) { *err = "gzip error: unknown compression method"; return 0; }
    const uint8_t flg = p[3];
    if (flg & 0xE0) { *err = "gzip error: unknown header flags set"; return 0; }
    p += 10;
    if (flg & 4) {                                         // FEXTRA
        if (end - p < 2) { *err = trunc; return 0; }
        const size_t xlen = (size_t)p[0] | ((size_t)p[1] << 8);
        p += 2;
        if ((size_t)(end - p) < xlen) { *err = trunc; return 0; }
        p += xlen;
    }
    for (int f = 0; f < 2; ++f) {                          // FNAME, FCOMMENT: zero-terminated
        if (flg & (f == 0 ? 8 : 16)) {
            const void *z = memchr(p, 0, (size_t)(end - p));
            if (!z) { *err = trunc; return 0; }
            p = (const uint8_t *)z + 1;
        }
    }
    if (flg & 2) {                                         // FHCRC
        if (end - p < 2) { *err = trunc; return 0; }
        p += 2;
    }
    return (size_t)(p - p0);
}

void GzipMemberDecoder::begin(const uint8_t *in, const uint8_t *in_end) {
    in_ = in; in_end_ = in_end;
    started_ = false;
    crc_ = 0;
    total_out_ = 0;
    err_ = "";
}

GzipMemberDecoder::Status GzipMemberDecoder::run(uint8_t *out, size_t cap, size_t *produced) {
    *produced = 0;
    if (!started_) {
        const size_t hdr = parse_header(in_, in_end_, &err_);
        if (!hdr) return kError;
        core_.begin(in_, in_end_, (uint64_t)hdr * 8u);
        started_ = true;
    }
    const DeflateCore<uint8_t>::Status st = core_.run(out, cap, produced);
    crc_ = crc32_update(crc_, out, *produced);
    total_out_ += *produced;
    if (st == DeflateCore<uint8_t>::kPieceFull) return kPieceFull;
    if (st != DeflateCore<uint8_t>::kStreamEnd) { err_ = core_.error(); return kError; }
    // trailer: CRC-32 and ISIZE at the next byte boundary
    const uint8_t *t = in_ + (core_.bit_pos() + 7) / 8;
    if (in_end_ - t < 8) { err_ = DeflateCore<uint8_t>::truncated(); return kError; }
    const uint32_t want_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    const uint32_t want_size = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
    if (want_crc != crc_) { err_ = "gzip error: incorrect data check"; return kError; }
    if (want_size != (uint32_t)total_out_) { err_ = "gzip error: incorrect length check"; return kError; }
    in_ = t + 8;
    return kMemberEnd;
}

}  // namespace vf

// ---- CRC-32 ---------------------------------------------------------------------------------------
// Carry-less-multiply folding (Gopal, Ozturk et al., "Fast CRC Computation for Generic Polynomials Using
// PCLMULQDQ Instruction", Intel 2009) for the bulk, zlib's table code for heads and tails.  zlib 1.2.11's
// own crc32 runs at ~1 GB/s and took a third of the decoder's time.
#include <immintrin.h>
#include <zlib.h>
namespace vf {
namespace {

// internal (pre- and post-inverted) CRC state over len bytes, len >= 64 and a multiple of 16
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_fold(const uint8_t *buf, size_t len, uint32_t crc) {
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {                                    // four lanes of 16 bytes folded 64 bytes ahead
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
        y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
        y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4);           // four lanes -> one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);               // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i *)poly);            // Barrett reduction to 32 bits
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

uint32_t crc32_tables(uint32_t crc, const uint8_t *buf, size_t len) {
    while (len) {
        const size_t n = len > (1u << 30) ? (1u << 30) : len;
        crc = (uint32_t)::crc32((uLong)crc, buf, (uInt)n);
        buf += n; len -= n;
    }
    return crc;
}

}  // namespace

size_t count_byte(const uint8_t *p, size_t n, uint8_t c) {
    size_t k = 0, i = 0;
    const __m128i cv = _mm_set1_epi8((char)c);
    for (; i + 64 <= n; i += 64) {
        const unsigned m0 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i)), cv));
        const unsigned m1 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i + 16)), cv));
        const unsigned m2 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i + 32)), cv));
        const unsigned m3 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i + 48)), cv));
        k += (size_t)__builtin_popcountll((uint64_t)m0 | ((uint64_t)m1 << 16) | ((uint64_t)m2 << 32) | ((uint64_t)m3 << 48));
    }
    for (; i < n; ++i) k += p[i] == c;
    return k;
}

uint32_t crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    return (uint32_t)::crc32_combine((uLong)crc_a, (uLong)crc_b, (z_off_t)len_b);
}

uint32_t crc32_update(uint32_t crc, const uint8_t *buf, size_t len) {
    static const bool has_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (has_clmul && len >= 64) {
        const size_t bulk = len & ~(size_t)15;
        crc = ~crc32_fold(buf, bulk, ~crc);
        buf += bulk; len -= bulk;
    }
    return crc32_tables(crc, buf, len);
}
}  // namespace vf
