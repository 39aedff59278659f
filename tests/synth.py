"""Seeded synthetic reads for parity tests (test infrastructure; small sizes).

Read = flank5 | prefix' | variant | suffix' | filler, following SURVEY.md 8(d): adapters carry
substitutions / indels at given rates, a share of reads carries no adapters, lengths may vary, and
optional 'N' / lowercase bytes exercise the byte-wise kernel.
"""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def rand_seq(rng, n):
    return BASES[rng.integers(0, 4, size=n)].tobytes()


def mutate(rng, s: bytes, sub: float, indel: float) -> bytes:
    out = bytearray()
    for ch in s:
        u = rng.random()
        if u < sub:
            out.append(int(BASES[rng.integers(0, 4)]))
        elif u < sub + indel / 2:
            continue                                   # deletion
        elif u < sub + indel:
            out.append(ch)
            out.append(int(BASES[rng.integers(0, 4)]))  # insertion
        else:
            out.append(ch)
    return bytes(out)


def make_reads(seed, n, adapters, L=150, sub=0.03, indel=0.01, no_adapter=0.05, n_variants=50,
               flank_max=8, vary_len=False, dirty=0.0, codon_frac=0.9):
    rng = np.random.default_rng(seed)
    pre, suf = (a.encode() if isinstance(a, str) else a for a in adapters)
    room = max(L - len(pre) - len(suf) - flank_max - 8, 6)
    lib = []
    for _ in range(n_variants):
        k = int(rng.integers(2, max(room // 3, 3)))
        ln = 3 * k if rng.random() < codon_frac else 3 * k + int(rng.integers(1, 3))
        lib.append(rand_seq(rng, min(ln, room)))
    w = 1.0 / np.arange(1, n_variants + 1) ** 1.1
    w /= w.sum()
    reads = []
    for _ in range(n):
        Lr = int(rng.integers(max(L // 2, 8), L + 1)) if vary_len else L
        if rng.random() < no_adapter:
            r = rand_seq(rng, Lr)
        else:
            v = lib[int(rng.choice(n_variants, p=w))]
            r = rand_seq(rng, int(rng.integers(0, flank_max + 1))) + mutate(rng, pre, sub, indel) + v + mutate(rng, suf, sub, indel)
            if len(r) < Lr:
                r += rand_seq(rng, Lr - len(r))
            r = r[:Lr]
        if dirty and rng.random() < dirty:
            b = bytearray(r)
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(0, len(b)))
                b[p] = ord("N") if rng.random() < 0.5 else ord(chr(b[p]).lower())
            r = bytes(b)
        reads.append(r)
    return reads


def write_fastq_gz(path, reads, members=1, crlf=False):
    import gzip
    nl = b"\r\n" if crlf else b"\n"
    per = max((len(reads) + members - 1) // members, 1)
    with open(path, "wb") as f:
        for m in range(members):
            chunk = reads[m * per:(m + 1) * per]
            if not chunk and m:
                continue
            body = b"".join(b"@r%d" % (m * per + i) + nl + r + nl + b"+" + nl + b"F" * len(r) + nl for i, r in enumerate(chunk))
            f.write(gzip.compress(body, compresslevel=1))


def fastq_text(reads, first=0, header_extra=b""):
    return b"".join(b"@r%d" % (first + i) + header_extra + b"\n" + r + b"\n+\n" + b"F" * len(r) + b"\n" for i, r in enumerate(reads))


def write_bgzf(path, text, block=65280):
    """BGZF (bgzip) container: gzip members of <= 64 KiB each carrying a 'BC' extra field, plus the
    28-byte empty EOF block"""
    import struct
    import zlib
    out = bytearray()
    for lo in list(range(0, len(text), block)) + [None]:
        data = b"" if lo is None else text[lo:lo + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(data) + co.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        out += b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
        out += body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(out)
