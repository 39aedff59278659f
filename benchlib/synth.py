"""ctypes front-end of the seeded synthetic read generator (benchlib/vfind_synth.h, libvfind_synth.so).
Bench / test tooling, not part of the product: reads are a pure function of (config, read index)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvfind_synth.so")
_L = None


def build() -> str:
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def row_dwords_for(max_len: int) -> int:
    """row width of the packed layout (include/vfind_hip.h: vf_row_dwords_for)"""
    return 10 if max_len <= 160 else 20 if max_len <= 320 else 40 if max_len <= 640 else 0


class _Cfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("read_len", C.c_uint32), ("flank_max", C.c_uint32),
                ("prefix", C.c_char_p), ("prefix_len", C.c_uint32), ("suffix", C.c_char_p), ("suffix_len", C.c_uint32),
                ("n_variants", C.c_uint64), ("zipf_s", C.c_double), ("sub_rate", C.c_double), ("indel_rate", C.c_double),
                ("no_adapter_frac", C.c_double), ("non_codon_frac", C.c_double)]


def _bind():
    global _L
    if _L is None:
        src = os.path.join(_HERE, "vf_synth.cpp")
        if not os.path.exists(_SO) or (os.path.exists(src) and os.path.getmtime(_SO) < os.path.getmtime(src)):
            build()
        _L = C.CDLL(_SO)
    L = _L
    if not getattr(L, "_synth_bound", False):
        L.vf_synth_adapters.restype = C.c_int
        L.vf_synth_adapters.argtypes = [C.c_uint64, C.c_uint32, C.c_char_p, C.c_char_p]
        L.vf_synth_create.restype = C.c_int
        L.vf_synth_create.argtypes = [C.POINTER(_Cfg), C.POINTER(C.c_void_p)]
        L.vf_synth_destroy.restype = None
        L.vf_synth_destroy.argtypes = [C.c_void_p]
        L.vf_synth_generate.restype = C.c_int
        L.vf_synth_generate.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L._synth_bound = True
    return L


def adapters(seed: int, m: int):
    L = _bind()
    a, b = C.create_string_buffer(m + 1), C.create_string_buffer(m + 1)
    if L.vf_synth_adapters(seed, m, a, b):
        raise ValueError("cannot draw adapters")
    return a.value.decode(), b.value.decode()


class Generator:
    """SURVEY.md 8(d) read model.  cfg-2 defaults: L=150, 0.5 % substitutions, 0.05 % indels."""

    def __init__(self, seed: int, adapters_, read_len=150, flank_max=8, n_variants=1_000_000, zipf_s=1.1,
                 sub_rate=0.005, indel_rate=0.0005, no_adapter_frac=0.05, non_codon_frac=0.02):
        self._L = _bind()
        self._p, self._s = adapters_[0].encode(), adapters_[1].encode()
        cfg = _Cfg(seed, read_len, flank_max, self._p, len(self._p), self._s, len(self._s), n_variants, zipf_s,
                   sub_rate, indel_rate, no_adapter_frac, non_codon_frac)
        self._g = C.c_void_p()
        if self._L.vf_synth_create(C.byref(cfg), C.byref(self._g)):
            raise ValueError("bad synth config")
        self.read_len = read_len
        self.row_dwords = row_dwords_for(read_len)

    def close(self):
        if self._g:
            self._L.vf_synth_destroy(self._g)
            self._g = None

    def __del__(self):
        self.close()

    def ascii(self, first: int, n: int, threads: int = 8):
        """-> (uint8 bytes [n*L], uint64 offsets [n+1])"""
        out = np.empty(n * self.read_len, np.uint8)
        if self._L.vf_synth_generate(self._g, first, n, out.ctypes.data, None, 0, None, threads):
            raise RuntimeError("synth failed")
        offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(self.read_len)
        return out, offs

    def packed_into(self, first: int, n: int, rows_ptr: int, lens_ptr: int, threads: int = 8):
        """the same into memory the caller owns (e.g. the library's page-locked batch staging): rows n * W uint32, lens n uint16"""
        if self._L.vf_synth_generate(self._g, first, n, None, rows_ptr, self.row_dwords, lens_ptr, threads):
            raise RuntimeError("synth failed")

    def packed(self, first: int, n: int, threads: int = 8):
        """-> (uint32 rows [n, W], uint16 lens [n])"""
        rows = np.empty((n, self.row_dwords), np.uint32)
        lens = np.empty(n, np.uint16)
        if self._L.vf_synth_generate(self._g, first, n, None, rows.ctypes.data, self.row_dwords, lens.ctypes.data, threads):
            raise RuntimeError("synth failed")
        return rows, lens
