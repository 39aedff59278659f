// vf_pinflate.h -- one gzip member decoded by several threads (for ordinary single-member .fastq.gz).
//
// A deflate stream is sequential only because each block may copy from the 32 KiB of text before it.
// Following Kerbiriou & Chikhi ("Parallel decompression of gzip-compressed files and random access to
// DNA sequences", 2019; pugz): the compressed stream is cut on a fixed grid; for every grid cell a worker
// searches the first bit position where a dynamic-Huffman block header parses into complete codes and
// its block decodes cleanly, and decodes from there to the first block boundary beyond the cell with
// the unknown window represented by markers (DeflateCore<uint16_t>).  The coordinator walks the cells in
// order: a cell's result is used only if its start is exactly the bit position the text before it ended
// on - then its markers are replaced from the now known window (in parallel again) - otherwise the
// coordinator decodes that stretch itself.  Wrong guesses cost time, never correctness; CRC-32 and ISIZE
// of the member are verified at the end as in the sequential decoder.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>

namespace vf {

struct ParallelInflateIO {
    size_t piece_bytes = 0;                                         // size of the buffers get_buffer hands out
    std::function<uint8_t *()> get_buffer;                          // a fresh output buffer (may block)
    std::function<bool(uint8_t *, size_t, size_t)> put_piece;       // (buffer, bytes, newlines) in stream order; false = abort
    std::function<void(uint8_t *)> drop_buffer;                     // an unused buffer goes back
    std::function<void(uint64_t)> progress;                         // compressed offset all text so far came from (optional)
};

// Decodes the member starting at `in`.  On success *member_end points just past its trailer.
// Returns false with *err set on a corrupt / truncated stream or when put_piece asked to abort ("aborted").
bool inflate_member_parallel(const uint8_t *in, const uint8_t *in_end, unsigned threads, const ParallelInflateIO &io,
                             const uint8_t **member_end, const char **err);

}  // namespace vf
