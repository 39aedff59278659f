"""ctypes front-end of oracle/libvf_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (vfind_amd/) never does; it fails loudly when its HIP library is missing.

Each function cites the reference line it checks (reference = vfind 0.3.1, src/lib.rs).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Iterable, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvf_oracle.so")


class _Params(C.Structure):
    _fields_ = [
        ("prefix", C.c_char_p), ("prefix_len", C.c_int32),
        ("suffix", C.c_char_p), ("suffix_len", C.c_int32),
        ("match", C.c_int32), ("mismatch", C.c_int32),
        ("gap_open", C.c_int32), ("gap_extend", C.c_int32),
        ("accept_prefix", C.c_double), ("accept_suffix", C.c_double),
        ("skip_translation", C.c_int32),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.vfo_find_exact.restype = C.c_int64
        L.vfo_find_exact.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]
        L.vfo_sg_stats.restype = None
        L.vfo_sg_stats.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32] + [C.c_int32] * 4 + [C.POINTER(C.c_int32)] * 4
        L.vfo_threshold.restype = C.c_int
        L.vfo_threshold.argtypes = [C.c_double, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        L.vfo_translate.restype = C.c_int64
        L.vfo_translate.argtypes = [C.c_char_p, C.c_int64, C.c_char_p]
        L.vfo_spans.restype = C.c_int
        L.vfo_spans.argtypes = [C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.vfo_count.restype = C.c_int
        L.vfo_count.argtypes = [C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
        L.vfo_count_reference_shape.restype = C.c_int
        L.vfo_count_reference_shape.argtypes = [C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
        L.vfo_count_fastq.restype = C.c_int
        L.vfo_count_fastq.argtypes = [C.POINTER(_Params), C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.vfo_table_free.argtypes = [C.c_void_p]
        L.vfo_table_size.restype = C.c_uint64
        L.vfo_table_size.argtypes = [C.c_void_p]
        L.vfo_table_key_bytes.restype = C.c_uint64
        L.vfo_table_key_bytes.argtypes = [C.c_void_p]
        L.vfo_table_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _b(x) -> bytes:
    return x.encode() if isinstance(x, str) else bytes(x)


def _params(adapters, match_score=3, mismatch_score=-2, gap_open_penalty=5, gap_extend_penalty=2,
            accept_prefix_alignment=0.75, accept_suffix_alignment=0.75, skip_translation=False):
    p, s = _b(adapters[0]), _b(adapters[1])
    P = _Params(p, len(p), s, len(s), match_score, mismatch_score, gap_open_penalty, gap_extend_penalty,
                accept_prefix_alignment, accept_suffix_alignment, int(bool(skip_translation)))
    P._keep = (p, s)
    return P


def pack_reads(reads: Iterable[bytes]):
    """list of byte strings -> (uint8 bytes, uint64 offsets[n+1])"""
    reads = [_b(r) for r in reads]
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        offs[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    data = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    return data, offs


def find_exact(seq, adapter) -> int:
    """memmem::find (lib.rs:148); -1 = None"""
    seq, adapter = _b(seq), _b(adapter)
    return lib().vfo_find_exact(seq, len(seq), adapter, len(adapter))


def sg_stats(adapter, seq, match=3, mismatch=-2, gap_open=5, gap_extend=2):
    """parasail sg_stats_scan restated (lib.rs:155-160): -> (score, length, end_query, end_ref)"""
    adapter, seq = _b(adapter), _b(seq)
    out = [C.c_int32() for _ in range(4)]
    lib().vfo_sg_stats(adapter, len(adapter), seq, len(seq), match, mismatch, gap_open, gap_extend,
                       *[C.byref(o) for o in out])
    return tuple(o.value for o in out)


def threshold(accept: float, match: int, m: int):
    """lib.rs:100-110 + :260-261 -> (skip_alignment, integer threshold T); ValueError as the reference"""
    skip, T = C.c_int32(), C.c_int64()
    if lib().vfo_threshold(accept, match, m, C.byref(skip), C.byref(T)):
        raise ValueError("Accept alignment threshold must be between 0 and 1.")
    return bool(skip.value), T.value


def translate(seq):
    """lib.rs:16-44; None when len % 3 != 0"""
    seq = _b(seq)
    out = C.create_string_buffer(len(seq) // 3 + 1)
    k = lib().vfo_translate(seq, len(seq), out)
    return None if k < 0 else out.raw[:k].decode("ascii")


def _check(rc: int, path: str = ""):
    if rc == 0:
        return
    if rc == 1:
        raise ValueError("Accept alignment threshold must be between 0 and 1.")
    if rc == 2:
        raise FileNotFoundError(path)
    raise RuntimeError({3: "input is not gzip", 4: "malformed FASTQ or truncated gzip"}.get(rc, f"oracle error {rc}"))


def spans(reads_or_packed, adapters, n_threads: int = 1, **kw):
    """per-read find_adapter_match results (lib.rs:278-286): (starts, ends) int64, -1 = None"""
    data, offs = reads_or_packed if isinstance(reads_or_packed, tuple) else pack_reads(reads_or_packed)
    n = len(offs) - 1
    P = _params(adapters, **kw)
    st = np.empty(n, np.int64)
    en = np.empty(n, np.int64)
    _check(lib().vfo_spans(C.byref(P), data.ctypes.data, offs.ctypes.data, n, st.ctypes.data, en.ctypes.data, n_threads))
    return st, en


def _dump(tab) -> dict:
    L = lib()
    n = L.vfo_table_size(tab)
    kb = L.vfo_table_key_bytes(tab)
    keys = np.zeros(max(kb, 1), np.uint8)
    offs = np.zeros(n + 1, np.uint64)
    cnts = np.zeros(max(n, 1), np.uint64)
    L.vfo_table_dump(tab, keys.ctypes.data, offs.ctypes.data, cnts.ctypes.data)
    L.vfo_table_free(tab)
    raw = keys.tobytes()
    return {raw[int(offs[i]):int(offs[i + 1])]: int(cnts[i]) for i in range(n)}


def count(reads_or_packed, adapters, n_threads: int = 1, **kw) -> dict:
    """work+reduce closures over a batch (lib.rs:275-306): {key bytes: count}"""
    data, offs = reads_or_packed if isinstance(reads_or_packed, tuple) else pack_reads(reads_or_packed)
    P = _params(adapters, **kw)
    tab = C.c_void_p()
    _check(lib().vfo_count(C.byref(P), data.ctypes.data, offs.ctypes.data, len(offs) - 1, n_threads, C.byref(tab)))
    return _dump(tab)


def count_reference_shape(reads_or_packed, adapters, n_workers: int = 3, **kw) -> dict:
    """the same table in the reference's threading shape (lib.rs:271-306): n_workers run the `work` closure on record sets,
    the calling thread runs the reduce closure on every record in input order (BASELINE.md 3's CPU baseline)"""
    data, offs = reads_or_packed if isinstance(reads_or_packed, tuple) else pack_reads(reads_or_packed)
    P = _params(adapters, **kw)
    tab = C.c_void_p()
    _check(lib().vfo_count_reference_shape(C.byref(P), data.ctypes.data, offs.ctypes.data, len(offs) - 1, n_workers, C.byref(tab)))
    return _dump(tab)


def count_fastq(path: str, adapters, n_threads: int = 1, **kw) -> dict:
    """find_variants on a gzip FASTQ (lib.rs:219-319): {key bytes: count}"""
    P = _params(adapters, **kw)
    tab = C.c_void_p()
    nr = C.c_int64()
    _check(lib().vfo_count_fastq(C.byref(P), os.fsencode(path), n_threads, C.byref(tab), C.byref(nr)), path)
    return _dump(tab)


def sorted_table(d: dict) -> Sequence[tuple]:
    """canonical order used for parity: count desc, sequence asc"""
    return sorted(((k, v) for k, v in d.items()), key=lambda kv: (-kv[1], kv[0]))
