"""vf::GzipMemberDecoder (vfind_amd/csrc/vf_inflate.cpp) against zlib: every block type, code shape and
member layout the FASTQ reader can meet, decoded into pieces of several sizes (so matches and stored
blocks cross piece boundaries and reach into the kept history), plus corrupted and truncated streams,
which must be reported - never crash (the harness is built with ASan + UBSan)."""
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("inflate") / "inflate_harness")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", out,
                           os.path.join(ROOT, "tests/cpp/inflate_harness.cpp"), os.path.join(ROOT, "vfind_amd/csrc/vf_inflate.cpp"),
                           os.path.join(ROOT, "vfind_amd/csrc/vf_pinflate.cpp"), "-lz", "-lpthread"])
    return out


def _run(harness, tmp_path, blob, piece, threads=0, cell=None):
    src, dst = tmp_path / "in.gz", tmp_path / "out.bin"
    src.write_bytes(blob)
    env = dict(os.environ, VFIND_PINFLATE_CELL=str(cell)) if cell else None
    p = subprocess.run([harness, str(src), str(dst), str(piece), "1", str(threads)], capture_output=True, text=True, timeout=120, env=env)
    return p.returncode, (dst.read_bytes() if p.returncode == 0 else b""), p.stderr


def _fastq(rng, n):
    q = b"FFFFFFFF:FFFFFF,FFFFFFFFFFFFF:FFFFFFFFFFFFFFFFFFFFF,FFFFF:FFFFFFFFFFFFFFFFF"
    out = []
    for i in range(n):
        s = bytes(b"ACGT"[x] for x in rng.integers(0, 4, 72))
        out.append(b"@read%d\n%s\n+\n%s\n" % (i, s, q[:72]))
    return b"".join(out)


def _member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


def _cases():
    rng = np.random.default_rng(7)
    text = _fastq(rng, 4000)
    rnd = rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes()
    cases = {
        "empty": _member(b""),
        "one_byte": _member(b"A"),
        "fastq_l1": _member(text, 1), "fastq_l6": _member(text, 6), "fastq_l9": _member(text, 9),
        "stored_l0": _member(text[:200_000], 0),                       # stored blocks (64 KiB each)
        "random_l6": _member(rnd, 6),                                   # incompressible: stored / near-flat codes
        "fixed_huffman": _member(text[:3000], 6, zlib.Z_FIXED),
        "huffman_only": _member(text[:100_000], 6, zlib.Z_HUFFMAN_ONLY),
        "rle": _member(b"A" * 100_000 + b"CG" * 50_000 + b"ACG" * 30_000 + text[:5000], 6, zlib.Z_RLE),
        "long_runs": _member(b"\0" * 1_000_000 + b"xyz" * 100_000, 9),      # distance 1 / 3, length 258 chains
        "far_matches": _member((rnd[:30_000] + text[:2000]) * 12, 9),     # distances near 32 KiB
        "many_members": b"".join(_member(text[i:i + 997], 1 + i % 9) for i in range(0, 60_000, 997)),
        "sync_flushed": None, "header_fields": None, "bgzf_like": None,
    }
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = []
    for i in range(0, 120_000, 7000):
        parts.append(c.compress(text[i:i + 7000]))
        parts.append(c.flush(zlib.Z_SYNC_FLUSH if i % 14000 else zlib.Z_FULL_FLUSH))     # empty stored blocks inside the stream
    cases["sync_flushed"] = (b"".join(parts) + c.flush(), text[:126_000])
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(text[:50_000]) + raw.flush()
    trailer = zlib.crc32(text[:50_000]).to_bytes(4, "little") + (50_000).to_bytes(4, "little")
    hdr = b"\x1f\x8b\x08" + bytes([4 | 8 | 16 | 2]) + b"\0\0\0\0\0\x03" + (6).to_bytes(2, "little") + b"BC\x02\x00\x99\x99" + b"name.fq\0" + b"a comment\0" + b"\x12\x34"
    cases["header_fields"] = (hdr + body + trailer, text[:50_000])
    cases["bgzf_like"] = b"".join(_member(text[i:i + 60_000], 6) for i in range(0, len(text), 60_000)) + _member(b"")
    out = {}
    for k, v in cases.items():
        out[k] = v if isinstance(v, tuple) else (v, gzip.decompress(v))
    return out


CASES = _cases()


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("piece", [1024, 4099, 70_000, 1 << 22])
def test_decodes_like_zlib(harness, tmp_path, name, piece):
    blob, want = CASES[name]
    rc, got, err = _run(harness, tmp_path, blob, piece)
    assert rc == 0, err
    assert got == want


def test_corrupt_and_truncated_streams_are_reported(harness, tmp_path):
    rng = np.random.default_rng(11)
    blob, want = CASES["fastq_l6"]
    n_err = 0
    for trial in range(120):
        b = bytearray(blob)
        kind = trial % 4
        if kind == 0:
            b[int(rng.integers(10, len(b) - 8))] ^= 1 << int(rng.integers(0, 8))            # bit flip in the deflate data
        elif kind == 1:
            b = b[: int(rng.integers(1, len(b) - 1))]                                        # truncation
        elif kind == 2:
            b[len(b) - 1 - int(rng.integers(0, 8))] ^= 0x40                                  # trailer (CRC / ISIZE)
        else:
            i = int(rng.integers(10, len(b) - 40))
            b[i:i + 16] = rng.integers(0, 256, 16, dtype=np.uint8).tobytes()               # a burst of garbage
        rc, got, err = _run(harness, tmp_path, bytes(b), 4099 if trial % 2 else 1 << 20)
        assert rc in (0, 3), err                       # 3 = reported; anything else = crash / sanitizer abort
        if rc == 0:
            assert got == want                         # e.g. a flipped bit in a header field that does not matter
        else:
            assert "gzip error" in err
            n_err += 1
    assert n_err > 100
    for bad in (b"", b"\x1f", b"\x1f\x8b\x08\x00", b"not a gzip file at all", b"\x1f\x8b\x07" + b"\0" * 20, b"\x1f\x8b\x08\xe0" + b"\0" * 20):
        rc, _, err = _run(harness, tmp_path, bad, 4096)
        assert (rc == 3 and "gzip error" in err) or (bad == b"" and rc == 0)


def test_crc32_matches_zlib(harness, tmp_path):
    # the member trailer check exercises crc32_update over pieces of every alignment and length
    rng = np.random.default_rng(5)
    for n in (1, 2, 15, 16, 17, 63, 64, 65, 255, 1000, 4097, 100_003):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for piece in (1024, 1031):
            rc, got, err = _run(harness, tmp_path, _member(data, 0), piece)
            assert rc == 0 and got == data, err


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("cell,threads", [(4096, 3), (20_000, 2), (1 << 20, 4)])
def test_parallel_member_decoder_like_zlib(harness, tmp_path, name, cell, threads):
    """inflate_member_parallel: marker-mode cells, chain check, marker resolution; cells far smaller than a
    deflate block (most claims fail and the coordinator decodes itself) up to larger than the input"""
    blob, want = CASES[name]
    rc, got, err = _run(harness, tmp_path, blob, 70_000, threads=threads, cell=cell)
    assert rc == 0, err
    assert got == want


def test_parallel_member_decoder_reports_corruption(harness, tmp_path):
    rng = np.random.default_rng(13)
    blob, want = CASES["fastq_l6"]
    n_err = 0
    for trial in range(60):
        b = bytearray(blob)
        kind = trial % 3
        if kind == 0:
            b[int(rng.integers(10, len(b) - 8))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            b = b[: int(rng.integers(1, len(b) - 1))]
        else:
            b[len(b) - 1 - int(rng.integers(0, 8))] ^= 0x40
        rc, got, err = _run(harness, tmp_path, bytes(b), 70_000, threads=3, cell=8192)
        assert rc in (0, 3), err
        if rc == 0:
            assert got == want
        else:
            assert "gzip error" in err
            n_err += 1
    assert n_err > 50
