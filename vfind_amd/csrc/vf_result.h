// vf_result.h -- the result frame built on the device (vf_result.hip) from an exported table
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "vf_internal.h"

namespace vf {
// refs / counts / key arena: an exported table in device memory (the blob of vf_table_export); fills `out` with the frame of
// lib.rs:312-319, rows ordered by count descending then sequence ascending, duplicate strings summed
int build_result_on_device(hipStream_t s, const uint64_t *d_refs, const uint64_t *d_counts, const uint32_t *d_arena, uint64_t n,
                           bool check_utf8, vf_result &out, std::string &err);
hipError_t rebase_refs(hipStream_t s, const uint64_t *d_refs, uint64_t n, uint64_t arena_base, uint64_t *d_out);
}
