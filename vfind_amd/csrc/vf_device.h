// vf_device.h -- structures shared by the host runtime (vf_host.cpp) and the gfx950 kernels
// (vf_k_*.hip).  Internal; the public boundary is include/vfind_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vf {

// ---- packed-read geometry -------------------------------------------------------------------
// A read is a row of W little-endian dwords, base k at bits [2k,2k+1]; A=0 C=1 G=2 T=3.
constexpr int kMaxFastAdapter = 32;   // rows of the DP kept in VGPRs; adapter fits one u64
constexpr int kMaxFastRead = 640;     // W = 40
constexpr int16_t kNone = -1;         // span coordinate: no match          (Option::None)
constexpr int16_t kPending = -2;      // span coordinate: waits for the DP kernel

// ---- packed DP cell value (see DESIGN.md "DP cell encoding") ---------------------------------
//   bits 31..13  score, biased by -mismatch*row      (signed)
//   bits 12..10  tie preference (diag 4 > F 3/2 > E 1/0), cleared/normalised after every max
//   bits  9..0   alignment length, biased by 32-row
constexpr int kScoreShift = 13;
constexpr int kPrefShift = 10;
constexpr uint32_t kPrefMask = 7u << kPrefShift;
constexpr uint32_t kLenMask = (1u << kPrefShift) - 1u;
constexpr int kLenBias = 32;
constexpr int kNegScore = -20000;
constexpr uint32_t kMaxGrid = 2048;     // every kernel launches at most this many workgroups

struct DpParams {
    uint64_t pat;        // adapter, 2 bits per base
    uint32_t m;          // adapter length (1..32)
    uint32_t eq[4];      // eq[c] bit i set <=> adapter base i == c
    int32_t T;           // accept iff score >= T
    int32_t mismatch;    // for un-biasing
    int32_t match, gopen, gext;   // plain scoring values (dp_coop)
    uint32_t coop_max;            // classes 3 / 4 of at most this many tasks run one task per wave (dp_coop); 0 = never
    int32_t e_open_k, e_ext_k, f_open_k, f_ext_k;  // packed additive constants
    uint32_t delta;      // (match - mismatch) << kScoreShift
    uint32_t is_prefix;
    uint32_t independent; // 1 = test hook: run the suffix DP even when the prefix failed (lib.rs:278-286)
    int32_t kerr;         // max unit edits of an acceptable alignment (pre-DP filter); <0 = none can pass
    int32_t kerr_lo;      // the same bound for a task whose best end column has unit edit distance 1; <0 = not tighter
    uint32_t eq_top[4];   // eq[c] moved to the top m bits, low 32-m bits set (Myers filter)
    uint32_t pv_init;     // top m bits set
    // block filter (k_bfilter): kerr_lo + 1 disjoint adapter blocks, block b = bases [blk_off[b], blk_off[b] + blk_len[b]);
    // nblk = 0: off (no tightened bound, a block shorter than 5 bases, or more than 4 blocks)
    uint32_t nblk;
    uint32_t blk_off[4], blk_len[4];
    // Hamming certificate (hamming_band_class): kerr_h[h], h = 1..7 = the unit-edit bound of OPTIMAL alignments of a task that
    // has a diagonal at Hamming distance h (max_unit_errors of max(T, match*m - h*(match - mismatch))), -1 where that is no
    // tighter than kerr; kerr_h[0] = 1 if any entry is set (0 = certificate off)
    int32_t kerr_h[8];
    // The pre-DP scan's metric (vf_k_front.hip, myers_scan): indel = 0 unit edit distance (Myers), threshold kthr = kerr; indel = 1 indel
    // distance (a mismatch = 2 units), threshold kthr = the most indel units an acceptable alignment can have (host: max_indel_units),
    // chosen where that is no larger than kerr.  eq_top's low 32-m bits are ones for the unit scan and zeros for the indel scan.
    // dl_lo / dl_h[h]: how many columns inside the cluster of detected columns the end columns of OPTIMAL alignments lie under the
    // tightened bounds kerr_lo / kerr_h[h] - kthr minus the tightened bound in the scan's own units (0 = no shrink)
    uint32_t indel;
    int32_t kthr;
    int32_t dl_lo;
    int32_t dl_h[8];
};

struct ScanParams {
    uint64_t pre_pat, suf_pat;
    uint32_t pre_len, suf_len;
    uint32_t align_pre, align_suf;   // 1 = semi-global fallback enabled for that adapter
    uint32_t independent;            // 1 = both adapters searched regardless of the other's outcome
};

// ---- generic (byte-wise) path ---------------------------------------------------------------
struct GenericParams {
    const uint8_t *pre, *suf;      // device copies of the ASCII adapters
    uint32_t pre_len, suf_len;
    uint32_t align_pre, align_suf;
    int64_t T_pre, T_suf;
    int32_t match, mismatch, open, ext;
    uint32_t skip_translation;
    uint32_t count_mode;           // 1 = insert keys, 0 = spans only
};

// ---- device count table ---------------------------------------------------------------------
// Open addressing, linear probing, one 64-byte slot per key (layout and protocol: vf_table_dev.h).
// Exported entries: ref = key_off_dwords<<24 | 1<<23 | kind<<16 | len  (len: bytes for kind 0, bases
// for kind 1), keys compacted into the blob arena.
constexpr uint64_t kRefValid = 1ull << 23;
enum { kKindBytes = 0, kKindPacked = 1 };

enum Counter : int {        // uint64 counters[] in device memory
    C_ARENA_TOP = 0,        // dwords of the long-key arena handed out
    C_ENTRIES = 1,
    C_ERROR = 2,            // bit0 arena full, bit1 table full, bit2 key too long, bit3 count log inconsistent, bit4 key cache entry without a slot, bit5 unmergeable byte key
    C_KEY_DWORDS = 3,       // total key dwords of all entries (size of an export)
    C_ACCEPTED = 4,
    C_COUNTED = 5,
    C_DP_PREFIX = 6,
    C_DP_SUFFIX = 7,
    C_EXPORT_N = 8,
    C_EXPORT_KEYS = 9,
    C_SLOW = 10,            // reads k_finish sent through the insert path (one-shot lookup missed)
    C_DBG0 = 11,            // diagnosis build only (make ablation): insert-path batches of a wave,
    C_DBG1 = 12,            //   trips of their claim/publish loop,
    C_DBG2 = 13,            //   claiming CAS issued,
    C_DBG3 = 14,            //   claiming CAS won
    C_DBG4 = 15,            //   reads counted in a workgroup's key cache (printed with VFIND_DEBUG_CLASSES=1)
    C_NUM = 16
};
constexpr int kTableCounters = 4;       // C_ARENA_TOP..C_KEY_DWORDS are reset with the table

constexpr int kCountWords = 2;          // dwords of count storage per slot: one 64-bit count, apart from the slot line
constexpr uint32_t kChunk = 1024;       // dwords of key arena a workgroup takes per refill (arena_alloc)
constexpr uint32_t kChunkRows = 21;     // chunk-state rows: 16 batch streams, the main stream, 4 resident lanes

struct Table {
    unsigned long long *chunk_state;  // [kChunkRows][kMaxGrid] workgroup arena chunks (see arena_alloc)
    uint32_t chunk_row;               // row of the launching stream
    uint4 *slots;                     // 4 x uint4 per slot
    unsigned long long *counts;       // [slots] 64-bit counts, apart from the slot lines (only k_log_apply / k_log_direct and the
                                      // rare count-carrying inserts of merge, growth and the byte-wise path write them)
    uint32_t *arena;                  // long keys only
    uint64_t *counters;
    uint32_t slot_mask;
    uint64_t arena_cap;               // dwords
    // Epoch tags (letter-key tables only; 0 = off): a slot is live only if its w0 carries the table's epoch (bits 53..56), and
    // every key dword of a live letter key carries the epoch in the top bits of its four bytes (letters are ASCII: those bits
    // are free) - key_tag = the pattern.  vf_table_clear then bumps the epoch instead of zeroing the slot lines (128 MB of the
    // 151 MB a clear of a 2^21-slot table wrote; 2 GB at 2^25 slots): stale lines read as free slots, stale key dwords as
    // "not written yet".  Every 15th clear zeroes for real (the epoch wraps).  vf_table_dev.h, "Epoch tags".
    uint32_t epoch, key_tag;
};
constexpr uint32_t kTagBits = 0x80808080u;
inline uint32_t key_tag_of(uint32_t epoch) {
    uint32_t t = 0;
    for (int j = 0; j < 4; ++j) if ((epoch >> j) & 1u) t |= 0x80u << (8 * j);
    return t;
}

// ---- deferred count log (one per in-flight batch) ----------------------------------------------
// k_finish never touches a count: an occurrence of a key that is already in the table is written down as the key's slot
// index and applied after the kernel (lib.rs:292-306: `*entry += 1` for every read, here once per batch and slot range).
//   one[wave][cap_one]      slot indices, one occurrence each, appended by the wave with coalesced stores
//   many[block][kLogMany]   (slot, count) pairs: what the workgroup's LDS aggregator and key cache hold at kernel end
//   tot[2][kLogMaxParts]    entries per slot part (slot >> shift), written by k_finish, read by scatter and apply
//   cur[2][kLogMaxParts]    scatter cursors
// k_log_scatter groups the entries by part (counting sort: totals -> part starts, a run per workgroup and part reserved
// with one atomic), k_log_apply gives every range of kLogRange slots to one workgroup, which counts its part's entries
// in LDS and adds what it has to the 64-bit counts, contiguous, once.  parts == 0: k_log_direct adds entry by entry
// (tables so large that a batch touches few of their ranges).
constexpr uint32_t kLogMany = 1024 + 256;   // pairs a workgroup can leave: aggregator + key cache entries (vf_table_dev.h)
#ifndef VF_LOG_MAX_PARTS
#define VF_LOG_MAX_PARTS 256
#endif
constexpr uint32_t kLogMaxParts = VF_LOG_MAX_PARTS;     // (a power of two, at most the workgroup size)
constexpr uint32_t kLogRange = 8192;        // slots per workgroup of k_log_apply (32 KiB of LDS counters)
struct CountLog {
    uint32_t *one, *n_one;
    uint2 *many;
    uint32_t *n_many;
    uint32_t *tot, *cur;
    uint32_t *one_by_part;            // scatter output, same sizes as one / many
    uint2 *many_by_part;
    uint64_t cap_total;               // entries `one` / `one_by_part` can hold
    uint32_t cap_one;                 // per wave segment
    uint32_t parts, shift;            // parts == 0: direct mode
    uint32_t grid;                    // workgroups of the k_finish launch that wrote the log
};
uint64_t count_log_entries(uint64_t n_reads, int W, bool dna, uint32_t *grid, uint32_t *cap_one);   // geometry of launch_finish for n reads
uint64_t count_log_entries_max(uint64_t n_reads, uint32_t *grid_max);                              // ... over every row width and key kind

// ---- launchers (vf_k_*.hip) ------------------------------------------------------------------
struct BatchLists {        // per in-flight batch; lists are segmented per workgroup of k_scan
    const uint32_t *rows;             // the batch itself: the filter and DP stages gather their tasks' rows and lengths from it
    const uint16_t *lens;
    uint32_t *listP, *listS;          // DP tasks per adapter (grid*seg entries each): read index | suffix-pending flag << 31 (vf_kernels_common.h, "Task entries")
    uint32_t *listW[4], *winW[4];     // filter output by class: 0 banded (+ first diagonal + kBandBias), 1..3 windowed (+ window a | b<<16)
    uint32_t *winW2;                  // second window of class-3 (two-window) tasks
    uint32_t *listF;                  // filter output: tasks only the full-width DP can take
    uint32_t *listN, *winN;           // block-filter output: narrow-band tasks (+ first diagonal + kBandBias)
    uint32_t *listB, *winB;           // Myers-filter output: wide-band tasks (class 6: one narrow cluster, any edits), same window word
    uint32_t *listR;                  // block-filter output: tasks it could not classify (the Myers filter takes them)
    uint32_t *cntP, *cntS, *cntW[4], *cntF, *cntN, *cntR, *cntB;   // kMaxGrid entries each
    uint32_t *tot;                    // [2][8]: tasks per class (0..3, 4 = full-width rest, 5 = narrow band, 6 = wide band) of the prefix / suffix pass
    uint32_t grid, seg;               // from scan_geometry(n, W)
};

// the class lists of one filter pass as the stage functions of k_bmfilter take them
struct ClassSet {
    uint32_t *listN, *winN;           // narrow band (block filter)
    uint32_t *listB, *winB;           // wide band (Myers filter)
    uint32_t *listW[4], *winW[4];     // 0 banded, 1..3 windowed
    uint32_t *winW2, *listF, *listR;  // second window, full width, rest (tasks the block filter left for the Myers pass)
};

void scan_geometry(uint64_t n, int W, uint32_t *grid, uint32_t *seg);
uint64_t scan_list_entries(uint64_t n);           // list entries a batch of n reads needs at any row width
hipError_t launch_scan(hipStream_t s, int W, const uint32_t *rows, const uint16_t *lens, uint64_t n,
                       const ScanParams &sp, short2 *spans, BatchLists bl);
// (rows and lengths of the batch: bl.rows, bl.lens)
hipError_t launch_bmfilter(hipStream_t s, int W, const DpParams &dp, short2 *spans, BatchLists bl, uint64_t *counters);
hipError_t launch_filter(hipStream_t s, int W, const DpParams &dp, short2 *spans, BatchLists bl, uint64_t *counters);
hipError_t launch_dp_classes(hipStream_t s, int W, const DpParams &dp, short2 *spans, BatchLists bl, uint64_t *counters);
hipError_t launch_dp(hipStream_t s, int W, const DpParams &dp, short2 *spans, BatchLists bl, uint64_t *counters);
// k_finish, then the kernels that apply its count log (same n, same table, same log, same stream)
hipError_t launch_finish(hipStream_t s, int W, const uint32_t *rows, const uint16_t *lens, uint64_t n,
                         const short2 *spans, bool skip_translation, Table t, CountLog log);
hipError_t launch_count_log(hipStream_t s, int W, bool skip_translation, uint64_t n, Table t, CountLog log);
hipError_t launch_generic(hipStream_t s, const uint8_t *bytes, const uint64_t *offs, uint64_t n,
                          const GenericParams &gp, int32_t *starts, int32_t *ends, int32_t *ws,
                          uint64_t ws_stride, uint32_t ws_rows, Table t);
hipError_t launch_generic_indexed(hipStream_t s, const uint8_t *text, const uint32_t *idx_start, const uint32_t *idx_len,
                                  const uint32_t *n_dev, const GenericParams &gp, int32_t *ws, uint64_t ws_stride, uint32_t ws_rows, Table t);
hipError_t launch_line_index(hipStream_t s, const uint8_t *text, uint64_t n, uint32_t *tile_cnt, uint32_t *tile_off, uint32_t *line_end,
                             uint32_t *info);
hipError_t launch_frame_pack(hipStream_t s, int W, const uint8_t *text, const uint32_t *line_end, uint32_t n_rec, uint32_t *rows,
                             uint16_t *lens, uint32_t *gen_start, uint32_t *gen_len, uint32_t *info, bool all_generic);
struct ClearRegions { void *p[4]; uint64_t bytes[4]; };   // 16-byte aligned starts, sizes multiples of 4
hipError_t launch_clear(hipStream_t s, const ClearRegions &cr);
hipError_t launch_peek(hipStream_t s, const void *d_src, void *h_dst, uint32_t n_dwords);   // <= 64 dwords, device -> page-locked host, no copy engine
// G = 0: the whole table; otherwise the entries owned by `owner` of G in the multi-GPU exchange
hipError_t launch_export(hipStream_t s, Table t, uint64_t *out_refs, uint64_t *out_counts, uint32_t *out_arena, uint32_t G, uint32_t owner);
hipError_t launch_owner_count(hipStream_t s, Table t, uint32_t G, uint64_t *out);      // out[2 G]: entries, key dwords per owner (zeroed by the caller)
hipError_t launch_rehash(hipStream_t s, Table from, Table to);
hipError_t launch_merge(hipStream_t s, Table t, const uint64_t *refs, const uint64_t *counts,
                        const uint32_t *arena, uint64_t n);

}  // namespace vf
