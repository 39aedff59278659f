/*
 * vfind_hip.h -- C ABI of the MI355X-native variant-extraction engine (libvfind_hip.so).
 *
 * This is the drop-in boundary for vFind's per-read hot path.  A host written in any language
 * (the reference's Rust/PyO3 host, the C++/Python host shipped here) binds exactly these entry
 * points; nothing in the signatures is a torch, HIP or C++ type.  Each entry point names the
 * reference interface it replaces (reference = nsbuitrago/vfind 0.3.1, src/lib.rs).
 *
 * Conventions
 *   - every function returns 0 on success or a VF_E* code; vf_last_error() returns the message
 *     of the last failure on the calling thread (no exceptions or panics cross the ABI);
 *   - a vf_ctx is owned by one submitting thread at a time;
 *   - "host" pointers are ordinary CPU memory, "device" pointers are HBM addresses on the
 *     context's GPU (hipMalloc / torch allocations are both fine);
 *   - coordinates are in bases, 0-based, as find_adapter_match returns them; -1 means None.
 */
#ifndef VFIND_HIP_H
#define VFIND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VF_ABI_VERSION 4

enum {
    VF_OK = 0,
    VF_EINVAL = 1,    /* bad argument; Python maps it to ValueError                     */
    VF_EIO = 2,       /* cannot open / read the input; Python maps it to OSError       */
    VF_EFORMAT = 3,   /* not gzip / malformed FASTQ; reference panics (lib.rs:308)     */
    VF_EDEVICE = 4,   /* HIP runtime failure                                           */
    VF_ECAPACITY = 5, /* device count table or key arena full                          */
    VF_ENOMEM = 6
};

/* Keyword arguments of find_variants (src/lib.rs:169-182, 219-232) plus device placement.
 * The library derives skip_*_alignment and the integer score thresholds exactly as
 * align_threshold_preflight (lib.rs:100-110) and lib.rs:260-261 do. */
typedef struct vf_params {
    const uint8_t *prefix;           /* adapters.0 bytes (lib.rs:240) */
    uint32_t prefix_len;
    const uint8_t *suffix;           /* adapters.1 bytes (lib.rs:250) */
    uint32_t suffix_len;
    int32_t match_score;             /* default 3  */
    int32_t mismatch_score;          /* default -2 */
    int32_t gap_open_penalty;        /* default 5  */
    int32_t gap_extend_penalty;      /* default 2  */
    double accept_prefix_alignment;  /* default 0.75; 1.0 = exact only */
    double accept_suffix_alignment;  /* default 0.75 */
    uint32_t n_threads;              /* host decode workers (lib.rs:178), default 3 */
    uint32_t queue_len;              /* batches in flight per GPU (lib.rs:179), default 2 */
    int32_t skip_translation;        /* lib.rs:180 */
    int32_t device;                  /* HIP device ordinal */
    uint64_t table_slots;            /* initial count-table slots (64 B + 8 B each), power of two; 0 = 2^24; grows x4 on demand */
    uint64_t arena_bytes;            /* initial arena for keys longer than 56 bytes; 0 = 1 GiB; grows x4 on demand */
    uint32_t batch_reads;            /* reads per device batch; 0 = 2^18 */
    uint32_t reserved;
} vf_params;

typedef struct vf_ctx vf_ctx;
typedef struct vf_result vf_result;

/* Counters and device timings of a context since creation / vf_stats_reset. */
typedef struct vf_stats {
    uint64_t reads;            /* reads submitted                                        */
    uint64_t reads_generic;    /* reads routed through the byte-wise (non-ACGT) kernel   */
    uint64_t dp_prefix;        /* semi-global alignments run for the prefix adapter      */
    uint64_t dp_suffix;        /* semi-global alignments run for the suffix adapter      */
    uint64_t accepted;         /* reads with start < end                                 */
    uint64_t counted;          /* reads that incremented a key (after the len%3 drop)    */
    uint64_t table_entries;    /* distinct keys on the device                            */
    uint64_t arena_dwords;     /* total key dwords held by the table                     */
    /* device time per kernel, milliseconds, summed over launches (HIP events on the stream the kernel
       is launched on; filled only while profiling is on).  dp_window = the DP launch that follows the filter (one
       kernel per adapter, every task class), dp_rest = the whole-read DP of scoring schemes that give the filter
       no bound */
    double ms_scan, ms_filter, ms_dp_window, ms_dp_rest, ms_finish, ms_generic;
    uint64_t launches_scan, launches_filter, launches_dp_window, launches_dp_rest, launches_finish, launches_generic;
    uint64_t insert_path;      /* counted reads whose one-shot table lookup missed (new, displaced or racing keys) */
    uint64_t arena_top;        /* dwords of the long-key arena handed out so far (keys longer than 56 bytes live there) */
    double ms_result_build;    /* wall time of the latest vf_finish: table export, device dedupe + sort, copy of the frame */
    uint64_t result_rows;      /* rows of that frame */
    uint64_t dbg[4];           /* diagnosis build only (insert-path batches, loop trips, CAS issued, CAS won); 0 in the product */
    double ms_count_log;       /* device time of the kernels that apply a batch's count log to the table (profiling on) */
    uint64_t launches_count_log;
    /* ABI 4 */
    uint64_t dup_rows;         /* rows the latest vf_finish / vf_finish_blobs folded into another row because their strings were
                                  equal: 0 for one context's own table unless the insert protocol let a key in twice (a stale view
                                  of a slot; summed, so the frame is right either way - but a non-zero value in a single-lane run is
                                  a protocol bug showing); the number of keys several blobs share for vf_finish_blobs */
    /* stages of vf_run_fastq since creation / vf_stats_reset, host steady_clock, milliseconds (lib.rs:233-234, 271-274 replaced):
       wall = whole calls; inflate = time inside the gzip decoder summed over the inflate threads (fq_inflate_threads of them;
       a single large member decoded by all threads counts its wall time once per thread); seq_wait = the in-order sequencer
       waiting for decoded text (inflate-bound when large); queue_full = the sequencer waiting for a free place in the chunk
       queue (device- or submit-bound when large); submit = inside the per-chunk submit summed over the submitting threads, of
       which slot_wait = waiting for a free batch slot and index_wait = waiting for the chunk's line index to come back from
       the device; drain = the final wait for the device */
    double fq_wall_ms, fq_inflate_ms, fq_seq_wait_ms, fq_queue_full_ms, fq_submit_ms, fq_slot_wait_ms, fq_index_wait_ms, fq_drain_ms;
    uint64_t fq_comp_bytes, fq_text_bytes, fq_inflate_threads, fq_chunks;
    double ms_h2d, ms_frame;   /* device time of the text copies and of the line-index + frame + pack kernels (profiling on) */
    uint64_t launches_h2d, launches_frame;
} vf_stats;

const char *vf_last_error(void);
int vf_abi_version(void);

/* Replaces the set-up half of find_variants (lib.rs:233-261): validates both thresholds
 * (prefix first) with the reference's message "Accept alignment threshold must be between 0
 * and 1.", builds the adapter parameter block and allocates the device state. */
int vf_ctx_create(const vf_params *params, vf_ctx **out);
void vf_ctx_destroy(vf_ctx *ctx);

/* Replaces File::open -> MultiGzDecoder -> Reader -> parallel_fastq(work, func)
 * (lib.rs:233-234, 271-308) for one gzip FASTQ file: decode on host threads, pack, stream to
 * HBM overlapped with the previous batch, run the device path.  May be called repeatedly. */
int vf_run_fastq(vf_ctx *ctx, const char *path);

/* Replaces the `work` + `func` closures (lib.rs:275-306) for one record set held in host
 * memory: n_reads sequences, read r = bytes[offsets[r] .. offsets[r+1]).  Asynchronous with
 * respect to the device; buffers may be reused on return. */
int vf_submit_ascii(vf_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_reads);

/* Same for a batch already 2-bit packed by the host (A=0 C=1 G=2 T=3, base k of a read at
 * bits [2k,2k+1] of its row, little-endian dwords, row_dwords in {10,20,40}); ACGT-only reads. */
int vf_submit_packed(vf_ctx *ctx, const uint32_t *rows, uint32_t row_dwords, const uint16_t *lens,
                     uint64_t n_reads);

/* Same without the staging copy: the caller packs straight into page-locked memory the library owns.  vf_batch_acquire
 * hands out the staging of a free batch slot (it blocks while all queue_len slots are in flight): rows = n_reads *
 * row_dwords dwords in the layout above, lens = n_reads lengths; vf_batch_submit enqueues the first n_reads of it
 * (hipMemcpyAsync + kernels on the slot's own stream, overlapped with the other slots) and returns at once;
 * vf_batch_release gives an acquired batch back unused.  The pointers are invalid after submit / release. */
typedef struct vf_batch vf_batch;
int vf_batch_acquire(vf_ctx *ctx, uint64_t n_reads, uint32_t row_dwords, vf_batch **batch, uint32_t **rows, uint16_t **lens);
int vf_batch_submit(vf_ctx *ctx, vf_batch *batch, uint64_t n_reads);
int vf_batch_release(vf_ctx *ctx, vf_batch *batch);

/* Same for a packed batch already resident in HBM (the kernel-only benchmark path).  d_spans
 * may be NULL; otherwise it receives n_reads {int16 start, int16 end} pairs (-1 = None). */
int vf_process_device_packed(vf_ctx *ctx, const void *d_rows, uint32_t row_dwords, const void *d_lens,
                             uint64_t n_reads, void *d_spans);

/* Test hook at the find_adapter_match boundary (lib.rs:141-166, called at :278-286): per-read
 * cut coordinates computed by the device path; nothing is counted. */
int vf_spans_ascii(vf_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_reads,
                   int32_t *starts, int32_t *ends);

/* Waits for all submitted work (the end of parallel_fastq, lib.rs:308). */
int vf_sync(vf_ctx *ctx);

/* Replaces the DataFrame build (lib.rs:312-319): the (sequence, count) table, rows sorted by
 * count descending then sequence ascending (the reference's order is unspecified). */
int vf_finish(vf_ctx *ctx, vf_result **out);
/* The same frame from several exported tables at once (vf_table_export / vf_table_partition_export blobs in this GPU's
 * memory, e.g. the tables of all GPUs gathered here): equal strings are summed by the device result build, nothing is
 * merged into this context's count table. */
int vf_finish_blobs(vf_ctx *ctx, const void *const *d_blobs, const uint64_t *nbytes, uint32_t n_blobs, vf_result **out);
uint64_t vf_result_rows(const vf_result *r);
const uint8_t *vf_result_key_bytes(const vf_result *r);   /* concatenated UTF-8 keys  */
const uint64_t *vf_result_key_offsets(const vf_result *r); /* rows+1 offsets           */
const uint64_t *vf_result_counts(const vf_result *r);      /* rows counts (UInt64)     */
void vf_result_free(vf_result *r);

/* Arrow C Data Interface export of the frame (https://arrow.apache.org/docs/format/CDataInterface.html):
 * a struct array {sequence: large_utf8, count: uint64} whose buffers the exported array owns from
 * then on (zero-copy hand-off to polars / pyarrow in place of lib.rs:312-319).  After a successful
 * export the vf_result is empty and must still be freed with vf_result_free. */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
    const char *format; const char *name; const char *metadata; int64_t flags; int64_t n_children;
    struct ArrowSchema **children; struct ArrowSchema *dictionary;
    void (*release)(struct ArrowSchema *); void *private_data;
};
struct ArrowArray {
    int64_t length; int64_t null_count; int64_t offset; int64_t n_buffers; int64_t n_children;
    const void **buffers; struct ArrowArray **children; struct ArrowArray *dictionary;
    void (*release)(struct ArrowArray *); void *private_data;
};
#endif
int vf_result_export_arrow(vf_result *r, struct ArrowArray *out_array, struct ArrowSchema *out_schema);

/* Empties the device count table (a new find_variants call on a reused context).  Asynchronous:
 * ordered after every batch already submitted and before everything submitted afterwards.  From the first call on the
 * context owns a second table of the same size: the call hands the empty one to the batches that follow and clears the
 * used one beside them (VFIND_TABLE_SPARE=0 in the environment: one table, cleared in place). */
int vf_table_clear(vf_ctx *ctx);

/* Multi-GPU exchange of the (key,count) table (the reduce of lib.rs:292-306 across GPUs).
 * vf_table_export_size gives the byte size of this GPU's compacted table; vf_table_export
 * writes it to device memory the caller owns (e.g. an RCCL send buffer); vf_table_merge adds a
 * peer's exported table (device memory, e.g. an RCCL receive buffer) into this GPU's table.
 * Both calls return when the library's stream is done with the buffer; the library does not know the caller's streams:
 * work the caller has queued on the buffer elsewhere (a fill, a copy, a collective) must have finished before the call. */
int vf_table_export_size(vf_ctx *ctx, uint64_t *nbytes);
int vf_table_export(vf_ctx *ctx, void *d_dst, uint64_t capacity, uint64_t *written);
int vf_table_merge(vf_ctx *ctx, const void *d_blob, uint64_t nbytes);   /* makes room first (grows the table / key arena) */
/* The rule above made enforceable: everything queued so far on `hip_stream` (a hipStream_t of the caller, NULL = the
 * null stream) becomes a dependency of whatever this context does next with device memory the caller hands it
 * (vf_table_export, vf_table_merge, vf_finish_blobs, vf_process_device_packed): an event recorded on the caller's stream,
 * waited for on the library's streams, no host synchronisation.  A fill or a collective still running on a buffer when
 * it was handed over once cost a two-rank rehearsal its counts (round 2, two_1.log). */
int vf_wait_stream(vf_ctx *ctx, void *hip_stream);

/* Owner-partitioned exchange: every key has one owner among `world` GPUs (a function of its hash).  partition_sizes gives
 * the byte size of each owner's share of this GPU's table (world values), partition_export writes one share in the format
 * of vf_table_export.  A host with its own transport sends share o to GPU o, lets every owner clear its table and merge
 * the shares it holds, and gathers the owners' tables (disjoint key sets) wherever it wants the sum. */
int vf_table_partition_sizes(vf_ctx *ctx, uint32_t world, uint64_t *nbytes);
int vf_table_partition_export(vf_ctx *ctx, uint32_t world, uint32_t owner, void *d_dst, uint64_t capacity, uint64_t *written);

/* The same exchange over RCCL (xGMI inside a node), one context per process and GPU: rank 0 draws a 128-byte unique id and
 * hands it to the other ranks by any means the host has; every rank calls vf_comm_init, and later vf_table_allreduce,
 * after which rank `root` holds the sum of all ranks' tables (the other ranks keep their own share).  librccl.so.1 is
 * loaded on first use. */
int vf_comm_unique_id(void *id128);
int vf_comm_init(vf_ctx *ctx, int rank, int world, const void *id128);
/* On a local failure of one rank (no memory for a buffer, a table that cannot grow) every rank returns an error: the ranks
 * exchange a status word before each round of transfers (its staging is allocated by vf_comm_init, so saying "I failed" cannot
 * itself fail), nobody is left waiting for a peer that has gone.  A failure of the transport itself, or a collective that does
 * not complete within VFIND_EXCHANGE_TIMEOUT_S seconds (default 120, 0 = wait for ever), closes an open ncclGroupStart,
 * aborts the communicator (ncclCommAbort) and returns an error: the call cannot hang on a dead peer.  The context needs a
 * new vf_comm_init after that. */
int vf_table_allreduce(vf_ctx *ctx, int root);

/* The same exchange between contexts of ONE process (one host thread per context, on one device or several): a group
 * object is the rendezvous, bytes move by device-to-device copies.  For a host that drives all GPUs of a node from one
 * process - and how the exchange code itself is run with world > 1 on a one-GPU box (RCCL refuses two ranks on one
 * device).  Every context of the group calls vf_table_allreduce from its own thread; the group outlives its contexts'
 * last call and is destroyed by the host. */
typedef struct vf_local_group vf_local_group;
int vf_comm_local_create(int world, vf_local_group **out);
void vf_comm_local_destroy(vf_local_group *group);
int vf_comm_init_local(vf_ctx *ctx, int rank, vf_local_group *group);

int vf_stats_get(vf_ctx *ctx, vf_stats *out);
int vf_stats_reset(vf_ctx *ctx);
/* on: 0 off; 1 HIP events around every kernel of the path; 2 around the two longest kernels (k_scan, k_finish) only; 3 the
   same on every third batch -- each bracket is two event packets on the stream: a step has about twenty of them with 1
   (measured 6 % of a step), two with 2 (3-5 %), two thirds of one with 3 */
int vf_set_profiling(vf_ctx *ctx, int on);

/* Host-side helpers shared by hosts that pack themselves. */
/* 2-bit packs one read; returns 0, or 1 when it holds a byte outside {A,C,G,T}. */
int vf_pack_read(const uint8_t *seq, uint32_t n, uint32_t *row, uint32_t row_dwords);
uint32_t vf_row_dwords_for(uint32_t max_read_len); /* 10, 20, 40, or 0 when > 640 */

#ifdef __cplusplus
}
#endif
#endif /* VFIND_HIP_H */
