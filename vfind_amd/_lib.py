"""ctypes binding of vfind_amd/libvfind_hip.so (the C ABI declared in include/vfind_hip.h).

There is no CPU fallback: if the HIP library is missing, or no MI355X is visible when a context
is created, the call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("VFIND_LIB_PATH") or os.path.join(_HERE, "libvfind_hip.so")      # (VFIND_LIB_PATH: another build of the library, A/B runs)

VF_OK, VF_EINVAL, VF_EIO, VF_EFORMAT, VF_EDEVICE, VF_ECAPACITY, VF_ENOMEM = range(7)
ABI_VERSION = 4          # VF_ABI_VERSION of include/vfind_hip.h


class VfParams(C.Structure):
    _fields_ = [
        ("prefix", C.c_char_p), ("prefix_len", C.c_uint32),
        ("suffix", C.c_char_p), ("suffix_len", C.c_uint32),
        ("match_score", C.c_int32), ("mismatch_score", C.c_int32),
        ("gap_open_penalty", C.c_int32), ("gap_extend_penalty", C.c_int32),
        ("accept_prefix_alignment", C.c_double), ("accept_suffix_alignment", C.c_double),
        ("n_threads", C.c_uint32), ("queue_len", C.c_uint32),
        ("skip_translation", C.c_int32), ("device", C.c_int32),
        ("table_slots", C.c_uint64), ("arena_bytes", C.c_uint64),
        ("batch_reads", C.c_uint32), ("reserved", C.c_uint32),
    ]


class VfStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "reads", "reads_generic", "dp_prefix", "dp_suffix", "accepted", "counted", "table_entries", "arena_dwords")] + [
        (n, C.c_double) for n in ("ms_scan", "ms_filter", "ms_dp_window", "ms_dp_rest", "ms_finish", "ms_generic")] + [
        (n, C.c_uint64) for n in ("launches_scan", "launches_filter", "launches_dp_window", "launches_dp_rest", "launches_finish", "launches_generic",
                                  "insert_path", "arena_top")] + [("ms_result_build", C.c_double), ("result_rows", C.c_uint64)] + [(f"dbg{k}", C.c_uint64) for k in range(4)] + [
        ("ms_count_log", C.c_double), ("launches_count_log", C.c_uint64), ("dup_rows", C.c_uint64)] + [
        (n, C.c_double) for n in ("fq_wall_ms", "fq_inflate_ms", "fq_seq_wait_ms", "fq_queue_full_ms", "fq_submit_ms", "fq_slot_wait_ms", "fq_index_wait_ms",
                                  "fq_drain_ms")] + [
        (n, C.c_uint64) for n in ("fq_comp_bytes", "fq_text_bytes", "fq_inflate_threads", "fq_chunks")] + [
        ("ms_h2d", C.c_double), ("ms_frame", C.c_double), ("launches_h2d", C.c_uint64), ("launches_frame", C.c_uint64)]


# every symbol include/vfind_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("vf_last_error", C.c_char_p, []),
    ("vf_abi_version", C.c_int, []),
    ("vf_ctx_create", C.c_int, [C.POINTER(VfParams), C.POINTER(_P)]),
    ("vf_ctx_destroy", None, [_P]),
    ("vf_run_fastq", C.c_int, [_P, C.c_char_p]),
    ("vf_submit_ascii", C.c_int, [_P, _P, _P, C.c_uint64]),
    ("vf_submit_packed", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint64]),
    ("vf_batch_acquire", C.c_int, [_P, C.c_uint64, C.c_uint32, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    ("vf_batch_submit", C.c_int, [_P, _P, C.c_uint64]),
    ("vf_batch_release", C.c_int, [_P, _P]),
    ("vf_process_device_packed", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint64, _P]),
    ("vf_spans_ascii", C.c_int, [_P, _P, _P, C.c_uint64, _P, _P]),
    ("vf_sync", C.c_int, [_P]),
    ("vf_finish", C.c_int, [_P, C.POINTER(_P)]),
    ("vf_finish_blobs", C.c_int, [_P, _P, _P, C.c_uint32, C.POINTER(_P)]),
    ("vf_result_rows", C.c_uint64, [_P]),
    ("vf_result_key_bytes", _P, [_P]),
    ("vf_result_key_offsets", _P, [_P]),
    ("vf_result_counts", _P, [_P]),
    ("vf_result_free", None, [_P]),
    ("vf_result_export_arrow", C.c_int, [_P, _P, _P]),
    ("vf_table_clear", C.c_int, [_P]),
    ("vf_table_export_size", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("vf_table_export", C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("vf_table_merge", C.c_int, [_P, _P, C.c_uint64]),
    ("vf_wait_stream", C.c_int, [_P, _P]),
    ("vf_table_partition_sizes", C.c_int, [_P, C.c_uint32, _P]),
    ("vf_table_partition_export", C.c_int, [_P, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("vf_comm_unique_id", C.c_int, [_P]),
    ("vf_comm_init", C.c_int, [_P, C.c_int, C.c_int, _P]),
    ("vf_table_allreduce", C.c_int, [_P, C.c_int]),
    ("vf_comm_local_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("vf_comm_local_destroy", None, [_P]),
    ("vf_comm_init_local", C.c_int, [_P, C.c_int, _P]),
    ("vf_stats_get", C.c_int, [_P, C.POINTER(VfStats)]),
    ("vf_stats_reset", C.c_int, [_P]),
    ("vf_set_profiling", C.c_int, [_P, C.c_int]),
    ("vf_pack_read", C.c_int, [_P, C.c_uint32, _P, C.c_uint32]),
    ("vf_row_dwords_for", C.c_uint32, [C.c_uint32]),
]


def build(force: bool = False) -> str:
    """compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)"""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "-s", "clean"])
    subprocess.check_call(["make", "-C", csrc, "-s"])
    return SO_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: build it with `make -C vfind_amd/csrc` (or __graft_entry__.build()). "
                "vfind_amd has no CPU fallback.")
        try:  # share one HIP runtime with torch when both are in the process (same SONAME)
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.vf_abi_version() != ABI_VERSION and not os.environ.get("VFIND_LIB_ANY_ABI"):      # (A/B runs against an older build)
            raise ImportError("libvfind_hip.so ABI version mismatch")
        _lib = L
    return _lib


class VfindError(RuntimeError):
    pass


def check(rc: int):
    if rc == VF_OK:
        return
    msg = (lib().vf_last_error() or b"").decode("utf-8", "replace")
    if rc == VF_EINVAL:
        raise ValueError(msg)
    if rc == VF_EIO:
        if "No such file" in msg:
            raise FileNotFoundError(msg)
        raise OSError(msg)
    if rc == VF_ENOMEM:
        raise MemoryError(msg)
    raise VfindError(msg)
