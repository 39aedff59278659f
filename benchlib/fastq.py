"""Synthetic gzip FASTQ files for the end-to-end measurements (bench / test tooling, not part of the product)."""
from __future__ import annotations

import gzip
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def write_fastq_gz(path: str, gen, n_reads: int, members: int = 16, level: int = 1, threads: int = 16, seed: int = 99, first: int = 0) -> int:
    """reads [first, first + n_reads) of the generator as a multi-member gzip FASTQ (members compressed in parallel); returns the file size"""
    per = (n_reads + members - 1) // members
    L = gen.read_len

    def one(m):
        lo, hi = m * per, min(n_reads, (m + 1) * per)
        if lo >= hi:
            return b""
        b, _ = gen.ascii(first + lo, hi - lo, 2)
        seqs = b.reshape(-1, L)
        n = hi - lo
        w = 12 + L + 3 + L + 1
        rec = np.empty((n, w), np.uint8)
        rec[:, 0] = ord("@")
        rec[:, 1] = ord("r")
        idx = first + lo + np.arange(n, dtype=np.int64)
        for d in range(9):
            rec[:, 2 + d] = ord("0") + (idx // 10 ** (8 - d)) % 10
        rec[:, 11] = ord("\n")
        rec[:, 12:12 + L] = seqs
        rec[:, 12 + L] = ord("\n")
        rec[:, 13 + L] = ord("+")
        rec[:, 14 + L] = ord("\n")
        rng = np.random.default_rng(seed + m)
        rec[:, 15 + L:15 + 2 * L] = rng.choice(np.frombuffer(b"FFFFFFFFFFFF::,#", np.uint8), (n, L))
        rec[:, w - 1] = ord("\n")
        return gzip.compress(rec.tobytes(), compresslevel=level)

    with ThreadPoolExecutor(max_workers=threads) as ex:
        blobs = list(ex.map(one, range(members)))
    with open(path, "wb") as f:
        for b in blobs:
            f.write(b)
    return os.path.getsize(path)
