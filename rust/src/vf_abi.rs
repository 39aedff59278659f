//! `extern "C"` declarations of every entry point of include/vfind_hip.h (VF_ABI_VERSION 4), in the header's order.
//! tests/test_abi.py checks that this list and the header name the same functions.
#![allow(dead_code)]
use std::ffi::{c_char, c_int, c_void};

#[repr(C)]
pub struct VfParams {
    pub prefix: *const u8,
    pub prefix_len: u32,
    pub suffix: *const u8,
    pub suffix_len: u32,
    pub match_score: i32,
    pub mismatch_score: i32,
    pub gap_open_penalty: i32,
    pub gap_extend_penalty: i32,
    pub accept_prefix_alignment: f64,
    pub accept_suffix_alignment: f64,
    pub n_threads: u32,
    pub queue_len: u32,
    pub skip_translation: i32,
    pub device: i32,
    pub table_slots: u64,
    pub arena_bytes: u64,
    pub batch_reads: u32,
    pub reserved: u32,
}
#[repr(C)]
pub struct VfCtx { _p: [u8; 0] }
#[repr(C)]
pub struct VfResult { _p: [u8; 0] }
#[repr(C)]
pub struct VfBatch { _p: [u8; 0] }

#[repr(C)]
pub struct VfStats {
    pub reads: u64, pub reads_generic: u64, pub dp_prefix: u64, pub dp_suffix: u64, pub accepted: u64, pub counted: u64,
    pub table_entries: u64, pub arena_dwords: u64,
    pub ms_scan: f64, pub ms_filter: f64, pub ms_dp_window: f64, pub ms_dp_rest: f64, pub ms_finish: f64, pub ms_generic: f64,
    pub launches_scan: u64, pub launches_filter: u64, pub launches_dp_window: u64, pub launches_dp_rest: u64, pub launches_finish: u64,
    pub launches_generic: u64, pub insert_path: u64, pub arena_top: u64,
    pub ms_result_build: f64, pub result_rows: u64, pub dbg: [u64; 4],
    pub ms_count_log: f64, pub launches_count_log: u64,
    pub dup_rows: u64,
    pub fq_wall_ms: f64, pub fq_inflate_ms: f64, pub fq_seq_wait_ms: f64, pub fq_queue_full_ms: f64, pub fq_submit_ms: f64, pub fq_slot_wait_ms: f64,
    pub fq_index_wait_ms: f64, pub fq_drain_ms: f64,
    pub fq_comp_bytes: u64, pub fq_text_bytes: u64, pub fq_inflate_threads: u64, pub fq_chunks: u64,
    pub ms_h2d: f64, pub ms_frame: f64, pub launches_h2d: u64, pub launches_frame: u64,
}

extern "C" {
    pub fn vf_last_error() -> *const c_char;
    pub fn vf_abi_version() -> c_int;
    pub fn vf_ctx_create(params: *const VfParams, out: *mut *mut VfCtx) -> c_int;
    pub fn vf_ctx_destroy(ctx: *mut VfCtx);
    pub fn vf_run_fastq(ctx: *mut VfCtx, path: *const c_char) -> c_int;
    pub fn vf_submit_ascii(ctx: *mut VfCtx, bytes: *const u8, offsets: *const u64, n_reads: u64) -> c_int;
    pub fn vf_submit_packed(ctx: *mut VfCtx, rows: *const u32, row_dwords: u32, lens: *const u16, n_reads: u64) -> c_int;
    pub fn vf_batch_acquire(ctx: *mut VfCtx, n_reads: u64, row_dwords: u32, batch: *mut *mut VfBatch, rows: *mut *mut u32, lens: *mut *mut u16) -> c_int;
    pub fn vf_batch_submit(ctx: *mut VfCtx, batch: *mut VfBatch, n_reads: u64) -> c_int;
    pub fn vf_batch_release(ctx: *mut VfCtx, batch: *mut VfBatch) -> c_int;
    pub fn vf_process_device_packed(ctx: *mut VfCtx, d_rows: *const c_void, row_dwords: u32, d_lens: *const c_void, n_reads: u64, d_spans: *mut c_void) -> c_int;
    pub fn vf_spans_ascii(ctx: *mut VfCtx, bytes: *const u8, offsets: *const u64, n_reads: u64, starts: *mut i32, ends: *mut i32) -> c_int;
    pub fn vf_sync(ctx: *mut VfCtx) -> c_int;
    pub fn vf_finish(ctx: *mut VfCtx, out: *mut *mut VfResult) -> c_int;
    pub fn vf_finish_blobs(ctx: *mut VfCtx, d_blobs: *const *const c_void, nbytes: *const u64, n_blobs: u32, out: *mut *mut VfResult) -> c_int;
    pub fn vf_result_rows(r: *const VfResult) -> u64;
    pub fn vf_result_key_bytes(r: *const VfResult) -> *const u8;
    pub fn vf_result_key_offsets(r: *const VfResult) -> *const u64;
    pub fn vf_result_counts(r: *const VfResult) -> *const u64;
    pub fn vf_result_free(r: *mut VfResult);
    pub fn vf_result_export_arrow(r: *mut VfResult, array: *mut polars_arrow::ffi::ArrowArray, schema: *mut polars_arrow::ffi::ArrowSchema) -> c_int;
    pub fn vf_table_clear(ctx: *mut VfCtx) -> c_int;
    pub fn vf_table_export_size(ctx: *mut VfCtx, nbytes: *mut u64) -> c_int;
    pub fn vf_table_export(ctx: *mut VfCtx, d_dst: *mut c_void, capacity: u64, written: *mut u64) -> c_int;
    pub fn vf_table_merge(ctx: *mut VfCtx, d_blob: *const c_void, nbytes: u64) -> c_int;
    pub fn vf_table_partition_sizes(ctx: *mut VfCtx, world: u32, nbytes: *mut u64) -> c_int;
    pub fn vf_table_partition_export(ctx: *mut VfCtx, world: u32, owner: u32, d_dst: *mut c_void, capacity: u64, written: *mut u64) -> c_int;
    pub fn vf_comm_unique_id(id128: *mut c_void) -> c_int;
    pub fn vf_comm_init(ctx: *mut VfCtx, rank: c_int, world: c_int, id128: *const c_void) -> c_int;
    pub fn vf_table_allreduce(ctx: *mut VfCtx, root: c_int) -> c_int;
    pub fn vf_wait_stream(ctx: *mut VfCtx, hip_stream: *mut c_void) -> c_int;
    pub fn vf_comm_local_create(world: c_int, out: *mut *mut c_void) -> c_int;
    pub fn vf_comm_local_destroy(group: *mut c_void);
    pub fn vf_comm_init_local(ctx: *mut VfCtx, rank: c_int, group: *mut c_void) -> c_int;
    pub fn vf_stats_get(ctx: *mut VfCtx, out: *mut VfStats) -> c_int;
    pub fn vf_stats_reset(ctx: *mut VfCtx) -> c_int;
    pub fn vf_set_profiling(ctx: *mut VfCtx, on: c_int) -> c_int;
    pub fn vf_pack_read(seq: *const u8, n: u32, row: *mut u32, row_dwords: u32) -> c_int;
    pub fn vf_row_dwords_for(max_read_len: u32) -> u32;
}
