// vf_inflate.h -- streaming gzip member decoder for the FASTQ input path (vf_fastq.cpp).
//
// Stands where flate2's MultiGzDecoder stands in the reference (src/lib.rs:233): RFC 1952 member
// framing around RFC 1951 deflate, CRC-32 and ISIZE verified.  Written for this reader's shape: the
// whole compressed file is mapped (no input refills), output goes to large caller-owned pieces, and a
// member may span many pieces (the last 32 KiB of text are kept as history between pieces).
// Tables: 11-bit primary literal/length table and 8-bit primary distance table with per-prefix
// subtables, 64-bit bit buffer refilled once per symbol group, word-wise match copies.
#pragma once
#include <cstddef>
#include <cstdint>

namespace vf {

class GzipMemberDecoder {
public:
    enum Status {
        kPieceFull = 0,      // the output piece is (nearly) full: hand it over and call again with a fresh one
        kMemberEnd = 1,      // one whole member decoded and verified; in_pos() is just past its trailer
        kError = 2,          // corrupt or truncated stream: message in error()
    };
    // start decoding a member at `in` (must stay mapped); `in_end` = end of the file
    void begin(const uint8_t *in, const uint8_t *in_end);
    // decode into out[0..cap); *produced = bytes written.  cap must be at least 1024.
    Status run(uint8_t *out, size_t cap, size_t *produced);
    const uint8_t *in_pos() const { return in_; }
    const char *error() const { return err_; }

private:
    enum State { kGzipHeader, kBlockHeader, kStored, kHuffman, kTrailer };
    static constexpr int kLitBits = 11, kDistBits = 8;
    static constexpr int kLitTableSize = (1 << kLitBits) + 1400, kDistTableSize = (1 << kDistBits) + 800;
    static constexpr size_t kWindow = 32768;

    bool parse_gzip_header();
    bool read_block_header();
    bool build(const uint8_t *lens, int nsyms, bool is_dist, uint32_t *table, int main_bits, int cap);
    bool finish_trailer();
    Status fail(const char *msg) { err_ = msg; return kError; }

    // byte-precise bit input for headers and stream ends
    bool need_bits(int n);

    const uint8_t *in_ = nullptr, *in_end_ = nullptr;
    uint64_t bitbuf_ = 0;
    int bitcnt_ = 0;
    State state_ = kGzipHeader;
    bool last_block_ = false;
    uint32_t stored_left_ = 0;
    uint32_t crc_ = 0;
    uint64_t total_out_ = 0;
    const char *err_ = "";
    size_t hist_len_ = 0;
    uint32_t lit_[kLitTableSize];
    uint32_t dist_[kDistTableSize];
    uint8_t hist_[kWindow];
};

// CRC-32 (gzip polynomial) of buf[0..len), continuing from crc
uint32_t crc32_update(uint32_t crc, const uint8_t *buf, size_t len);

}  // namespace vf
