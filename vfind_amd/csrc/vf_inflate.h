// vf_inflate.h -- deflate / gzip decoding for the FASTQ input path (vf_fastq.cpp).
//
// Stands where flate2's MultiGzDecoder stands in the reference (src/lib.rs:233): RFC 1952 member
// framing around RFC 1951 deflate, CRC-32 and ISIZE verified.  Written for this reader's shape: the
// whole compressed file is mapped (no input refills), output goes to large caller-owned pieces, and a
// stream may span many pieces (the last 32 KiB of text are kept as history between pieces).
// Tables: 11-bit primary literal/length table and 8-bit primary distance table with per-prefix
// subtables, 64-bit bit buffer refilled once per symbol group, word-wise match copies.
//
// DeflateCore<T> decodes blocks from any block boundary (bit position).  With T = uint8_t the text that
// precedes the start must be known (set_history).  With T = uint16_t it need not be: the window is
// filled with markers (256 + window index) and every symbol copied from it stays a marker, to be
// replaced once the real window is known - that is what lets one long deflate stream be decoded by
// several threads (vf_fastq.cpp, after Kerbiriou & Chikhi's pugz).
#pragma once
#include <cstddef>
#include <cstdint>

namespace vf {

constexpr size_t kDeflateWindow = 32768;
constexpr uint16_t kMarkerBase = 256;            // uint16_t symbols >= this are window references

template <typename T>
class DeflateCore {
public:
    enum Status {
        kPieceFull = 0,      // the output piece is (nearly) full: call again with a fresh one
        kBoundary = 1,       // stopped in front of a block header at or beyond the stop bit
        kStreamEnd = 2,      // the final block has ended; bit_pos() is just past it
        kError = 3,          // corrupt or truncated stream: message in error()
    };
    // start in front of a block header at absolute bit position bit_pos of the mapped file [base, in_end)
    void begin(const uint8_t *base, const uint8_t *in_end, uint64_t bit_pos);
    void set_stop_bit(uint64_t b) { stop_bit_ = b; }
    void set_history(const T *text, size_t n);                 // the n <= 32768 elements in front of the output
    void fill_markers();                                        // T = uint16_t: unknown window
    Status run(T *out, size_t cap, size_t *produced);          // cap >= 1024
    uint64_t bit_pos() const { return (uint64_t)(in_ - base_) * 8u - (uint64_t)bitcnt_; }
    const T *history(size_t *n) const { *n = hist_len_; return hist_; }
    const char *error() const { return err_; }
    static const char *truncated() { return "gzip error: truncated stream"; }

private:
    enum State { kBlockHeader, kStored, kHuffman, kDone };
    static constexpr int kLitBits = 11, kDistBits = 8;
    static constexpr int kLitTableSize = (1 << kLitBits) + 1400, kDistTableSize = (1 << kDistBits) + 800;

    bool read_block_header();
    bool build(const uint8_t *lens, int nsyms, bool is_dist, uint32_t *table, int main_bits, int cap);
    bool need_bits(int n);
    void keep_history(const T *out, size_t pos);

    const uint8_t *base_ = nullptr, *in_ = nullptr, *in_end_ = nullptr;
    uint64_t bitbuf_ = 0;
    int bitcnt_ = 0;
    uint64_t stop_bit_ = ~0ull;
    State state_ = kBlockHeader;
    bool last_block_ = false;
    uint32_t stored_left_ = 0;
    const char *err_ = "";
    size_t hist_len_ = 0;
    uint32_t lit_[kLitTableSize];
    uint32_t dist_[kDistTableSize];
    T hist_[kDeflateWindow];
};

class GzipMemberDecoder {
public:
    enum Status { kPieceFull = 0, kMemberEnd = 1, kError = 2 };
    // start decoding a member at `in` (must stay mapped); `in_end` = end of the file
    void begin(const uint8_t *in, const uint8_t *in_end);
    // decode into out[0..cap); *produced = bytes written.  cap must be at least 1024.
    Status run(uint8_t *out, size_t cap, size_t *produced);
    const uint8_t *in_pos() const { return in_; }
    const char *error() const { return err_; }
    // RFC 1952 header at p: returns the offset of the deflate data, or 0 with *err set
    static size_t parse_header(const uint8_t *p, const uint8_t *end, const char **err);

private:
    const uint8_t *in_ = nullptr, *in_end_ = nullptr;
    bool started_ = false;
    uint32_t crc_ = 0;
    uint64_t total_out_ = 0;
    const char *err_ = "";
    DeflateCore<uint8_t> core_;
};

// occurrences of byte c in p[0..n) (SSE2: 64 bytes a step)
size_t count_byte(const uint8_t *p, size_t n, uint8_t c);
// CRC-32 (gzip polynomial) of buf[0..len), continuing from crc
uint32_t crc32_update(uint32_t crc, const uint8_t *buf, size_t len);
// CRC-32 of A followed by B from crc(A), crc(B) and len(B)
uint32_t crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);

}  // namespace vf
