/* prints the layout of the ABI's structs as the C compiler sees them in include/vfind_hip.h (tests/test_abi.py compares
 * the ctypes mirror and the Rust declarations with it): one line per field, "struct field offset size" */
#include <stddef.h>
#include <stdio.h>
#include "vfind_hip.h"
#define F(S, f) printf(#S " " #f " %zu %zu\n", offsetof(S, f), sizeof(((S *)0)->f))
int main(void) {
    printf("vf_params sizeof %zu 0\n", sizeof(vf_params));
    F(vf_params, prefix); F(vf_params, prefix_len); F(vf_params, suffix); F(vf_params, suffix_len);
    F(vf_params, match_score); F(vf_params, mismatch_score); F(vf_params, gap_open_penalty); F(vf_params, gap_extend_penalty);
    F(vf_params, accept_prefix_alignment); F(vf_params, accept_suffix_alignment);
    F(vf_params, n_threads); F(vf_params, queue_len); F(vf_params, skip_translation); F(vf_params, device);
    F(vf_params, table_slots); F(vf_params, arena_bytes); F(vf_params, batch_reads); F(vf_params, reserved);
    printf("vf_stats sizeof %zu 0\n", sizeof(vf_stats));
    F(vf_stats, reads); F(vf_stats, reads_generic); F(vf_stats, dp_prefix); F(vf_stats, dp_suffix); F(vf_stats, accepted);
    F(vf_stats, counted); F(vf_stats, table_entries); F(vf_stats, arena_dwords);
    F(vf_stats, ms_scan); F(vf_stats, ms_filter); F(vf_stats, ms_dp_window); F(vf_stats, ms_dp_rest); F(vf_stats, ms_finish); F(vf_stats, ms_generic);
    F(vf_stats, launches_scan); F(vf_stats, launches_filter); F(vf_stats, launches_dp_window); F(vf_stats, launches_dp_rest);
    F(vf_stats, launches_finish); F(vf_stats, launches_generic); F(vf_stats, insert_path); F(vf_stats, arena_top);
    F(vf_stats, ms_result_build); F(vf_stats, result_rows); F(vf_stats, dbg); F(vf_stats, ms_count_log); F(vf_stats, launches_count_log);
    F(vf_stats, dup_rows);
    F(vf_stats, fq_wall_ms); F(vf_stats, fq_inflate_ms); F(vf_stats, fq_seq_wait_ms); F(vf_stats, fq_queue_full_ms); F(vf_stats, fq_submit_ms);
    F(vf_stats, fq_slot_wait_ms); F(vf_stats, fq_index_wait_ms); F(vf_stats, fq_drain_ms);
    F(vf_stats, fq_comp_bytes); F(vf_stats, fq_text_bytes); F(vf_stats, fq_inflate_threads); F(vf_stats, fq_chunks);
    F(vf_stats, ms_h2d); F(vf_stats, ms_frame); F(vf_stats, launches_h2d); F(vf_stats, launches_frame);
    return 0;
}
