"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/vfind_hip.h declares, validates thresholds like the reference (lib.rs:100-110) and fails
loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests.conftest import ROOT


def _lib():
    from vfind_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    return _lib


def test_header_symbols_all_exported():
    L = _lib()
    hdr = open(os.path.join(ROOT, "include", "vfind_hip.h")).read()
    declared = set(re.findall(r"\b(vf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vf_params", "vf_ctx", "vf_result", "vf_stats"}
    assert len(declared) >= 25
    so = C.CDLL(L.SO_PATH)
    for name in sorted(declared):
        assert hasattr(so, name), f"{name} declared in include/vfind_hip.h but not exported"
    assert {s[0] for s in L.SYMBOLS} == declared
    assert L.lib().vf_abi_version() == L.ABI_VERSION == int(re.search(r"#define VF_ABI_VERSION (\d+)", hdr).group(1))


def test_pack_read_layout():
    L = _lib().lib()
    seq = b"ACGTTGCA" * 5          # 40 bases
    row = np.zeros(10, np.uint32)
    assert L.vf_pack_read(seq, len(seq), row.ctypes.data, 10) == 0
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    for k, ch in enumerate(seq):
        assert (int(row[k // 16]) >> (2 * (k % 16))) & 3 == code[ch]
    assert row[3:].sum() == 0
    assert L.vf_pack_read(b"ACGN", 4, row.ctypes.data, 10) == 1
    assert L.vf_pack_read(b"acgt", 4, row.ctypes.data, 10) == 1
    assert [L.vf_row_dwords_for(n) for n in (0, 150, 160, 161, 320, 321, 640, 641)] == [10, 10, 10, 20, 20, 40, 40, 0]


@pytest.mark.parametrize("bad", [0.0, -0.5, 1.5, float("nan")])
def test_threshold_validation_matches_reference(bad):
    from vfind_amd import Engine
    with pytest.raises(ValueError, match="Accept alignment threshold must be between 0 and 1."):
        Engine(("ACGT", "TGCA"), accept_prefix_alignment=bad)
    with pytest.raises(ValueError, match="Accept alignment threshold must be between 0 and 1."):
        Engine(("ACGT", "TGCA"), accept_suffix_alignment=bad)


def test_missing_file_and_bad_args():
    import vfind_amd
    with pytest.raises(FileNotFoundError):
        vfind_amd.find_variants("/nonexistent/x.fq.gz", ("ACGT", "TGCA"))
    with pytest.raises(TypeError):
        vfind_amd.Engine(("ACGT",))
    with pytest.raises(OverflowError):
        vfind_amd.Engine(("ACGT", "TGCA"), n_threads=-1)


def test_no_silent_cpu_fallback():
    """without a GPU the product must refuse to run rather than compute on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vfind_amd import Engine, VfindError
    with pytest.raises((VfindError, RuntimeError)):
        Engine(("ACGT", "TGCA"))


def test_product_does_not_reference_oracle():
    for d, _, files in os.walk(os.path.join(ROOT, "vfind_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(d, f), errors="replace").read()
                assert "oracle" not in txt.replace("no CPU fallback", ""), f"{f} mentions the oracle"


def test_pack_read_matches_a_plain_packer():
    """vf_pack_read (SIMD on x86 with SSSE3 + BMI2, table loop otherwise): 2 bits per base, A0 C1 G2 T3, base k at
    bits [2k, 2k+1] of the row, zero padded; any byte outside ACGT is reported (the read then takes the
    byte-wise path)"""
    import ctypes
    import numpy as np
    from vfind_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    for t in range(4000):
        n = int(rng.integers(0, 161))
        s = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
        dirty = t % 5 == 0 and n > 0
        if dirty:
            s = s.copy()
            s[rng.integers(0, n)] = rng.choice(np.frombuffer(b"NacgtRn.@U\x00\xff", np.uint8))
        row = np.full(10, 0xDEADBEEF, np.uint32)
        rc = L.vf_pack_read(s.ctypes.data if n else None, n, row.ctypes.data, 10)
        if dirty:
            assert rc == 1
            continue
        want = np.zeros(10, np.uint32)
        for i, c in enumerate(s):
            want[i // 16] |= np.uint32(code[int(c)] << (2 * (i % 16)))
        assert rc == 0 and (row == want).all(), (t, n)
    assert L.vf_pack_read(np.frombuffer(b"A" * 161, np.uint8).ctypes.data, 161, np.zeros(10, np.uint32).ctypes.data, 10) == 1


def test_rust_shim_declares_the_same_entry_points():
    """rust/ is source only here (no rustc): its extern "C" block must at least name exactly the functions the header declares,
    and its vf_stats mirror must have as many 8-byte fields as the C struct"""
    hdr = open(os.path.join(ROOT, "include", "vfind_hip.h")).read()
    declared = set(re.findall(r"\b(vf_[a-z0-9_]+)\s*\(", hdr)) - {"vf_params", "vf_ctx", "vf_result", "vf_stats", "vf_batch"}
    rs = open(os.path.join(ROOT, "rust", "src", "vf_abi.rs")).read()
    assert set(re.findall(r"pub fn (vf_[a-z0-9_]+)\(", rs)) == declared
    from vfind_amd import _lib
    import ctypes as C
    body = rs[rs.index("pub struct VfStats"):rs.index('extern "C" {')]
    n_fields = len(re.findall(r"pub [a-z0-9_]+: (?:u64|f64)", body)) + 4 * len(re.findall(r"\[u64; 4\]", body))
    assert n_fields * 8 == C.sizeof(_lib.VfStats)


def test_exception_barrier_returns_a_code(monkeypatch):
    """include/vfind_hip.h: no exception crosses the ABI.  With the test hook that makes vf_finish throw, the entry point
    comes back with an error code and a message (no GPU is needed to get that far) - the process is still here to assert it"""
    from vfind_amd import _lib
    L = _lib.lib()
    out = C.c_void_p()
    monkeypatch.setenv("VFIND_TEST_THROW", "finish")
    assert L.vf_finish(None, C.byref(out)) == _lib.VF_ENOMEM
    assert b"out of host memory" in L.vf_last_error()
    monkeypatch.setenv("VFIND_TEST_THROW", "finish_length")
    assert L.vf_finish(None, C.byref(out)) == _lib.VF_EDEVICE
    assert b"internal error" in L.vf_last_error()
    monkeypatch.delenv("VFIND_TEST_THROW")
    assert L.vf_finish(None, C.byref(out)) == _lib.VF_EINVAL


def test_every_entry_point_sits_behind_the_barrier():
    """each `extern "C" int vf_*` of the library's sources is a one-line wrapper around vf_guard"""
    n = 0
    for f in ("vf_host.cpp", "vf_fastq.cpp", "vf_arrow.cpp"):
        src = open(os.path.join(ROOT, "vfind_amd", "csrc", f)).read()
        for m in re.finditer(r'^extern "C" int (vf_\w+)\(.*$', src, re.M):
            if m.group(1) == "vf_abi_version":
                continue
            assert "vf_guard(" in m.group(0), m.group(0)
            n += 1
    assert n >= 25


def _c_layout(tmp_path):
    """{struct: {field: (offset, size)}} as gcc lays out include/vfind_hip.h (tests/tools/abi_layout.c)"""
    import subprocess
    exe = str(tmp_path / "abi_layout")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "tools", "abi_layout.c"), "-o", exe])
    out = {}
    for line in subprocess.check_output([exe], text=True).splitlines():
        st, f, off, size = line.split()
        out.setdefault(st, {})[f] = (int(off), int(size))
    return out


def test_ctypes_mirror_has_the_c_layout(tmp_path):
    """size and every field offset of vf_params and vf_stats: the ctypes mirror against the C compiler's view of the header"""
    from vfind_amd import _lib
    lay = _c_layout(tmp_path)
    for cname, mirror in (("vf_params", _lib.VfParams), ("vf_stats", _lib.VfStats)):
        c = dict(lay[cname])
        assert c.pop("sizeof")[0] == C.sizeof(mirror), cname
        got = {}
        for name, _typ in mirror._fields_:
            d = getattr(mirror, name)
            base = re.sub(r"\d+$", "", name) if name.startswith("dbg") else name      # dbg0..dbg3 mirror uint64_t dbg[4]
            if base == "dbg":
                o, sz = got.get("dbg", (d.offset, 0))
                got["dbg"] = (min(o, d.offset), sz + d.size)
            else:
                got[name] = (d.offset, d.size)
        assert got == c, (cname, {k: (got.get(k), c.get(k)) for k in set(got) | set(c) if got.get(k) != c.get(k)})


def test_rust_structs_have_the_c_layout(tmp_path):
    """rust/src/vf_abi.rs cannot be compiled here (no rustc): its #[repr(C)] VfParams / VfStats are parsed and laid out by the
    C rules (natural alignment) and compared, field by field and in order, with the C compiler's view of the header"""
    lay = _c_layout(tmp_path)
    rs = open(os.path.join(ROOT, "rust", "src", "vf_abi.rs")).read()
    sizes = {"u8": 1, "u32": 4, "i32": 4, "u64": 8, "f64": 8, "c_int": 4}
    for rname, cname in (("VfParams", "vf_params"), ("VfStats", "vf_stats")):
        m = re.search(r"#\[repr\(C\)\]\s*pub struct %s \{(.*?)\n\}" % rname, rs, re.S)
        assert m, rname
        at, fields, max_align = 0, {}, 1
        for name, typ in re.findall(r"pub (\w+):\s*([^,\n]+?)\s*,", m.group(1)):
            arr = re.match(r"\[(\w+);\s*(\d+)\]", typ)
            if typ.startswith("*"):
                size = align = 8
            elif arr:
                align = sizes[arr.group(1)]
                size = align * int(arr.group(2))
            else:
                size = align = sizes[typ]
            at = (at + align - 1) // align * align
            fields[name] = (at, size)
            at += size
            max_align = max(max_align, align)
        total = (at + max_align - 1) // max_align * max_align
        c = dict(lay[cname])
        assert c.pop("sizeof")[0] == total, rname
        assert list(fields.items()) == sorted(c.items(), key=lambda kv: kv[1][0]), rname


def test_rust_check_and_python_check_map_the_same_codes():
    """rust/src/lib.rs::check and vfind_amd/_lib.py::check turn the same VF_E* codes into the same exception families
    (INTEGRATION.md section 2): ValueError, OSError, MemoryError, everything else a runtime error"""
    from vfind_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vfind_hip.h")).read()
    codes = {name: int(v) for name, v in re.findall(r"\b(VF_E[A-Z]+)\s*=\s*(\d+)", hdr)}
    assert codes == {"VF_EINVAL": 1, "VF_EIO": 2, "VF_EFORMAT": 3, "VF_EDEVICE": 4, "VF_ECAPACITY": 5, "VF_ENOMEM": 6}
    rs = open(os.path.join(ROOT, "rust", "src", "lib.rs")).read()
    body = rs[rs.index("fn check("):rs.index("struct Ctx")]
    arms = {int(c): exc for c, exc in re.findall(r"(\d+)\s*=>\s*(Py\w+)::new_err", body)}
    default = re.search(r"_\s*=>\s*(Py\w+)::new_err", body).group(1)
    family = {"PyValueError": ValueError, "PyOSError": OSError, "PyMemoryError": MemoryError, "PyRuntimeError": RuntimeError}

    class _L:                                    # _lib.check only asks the library for the message
        @staticmethod
        def vf_last_error():
            return b"No such file or directory"
    saved = _lib._lib
    _lib._lib = _L
    try:
        for name, code in codes.items():
            want = family[arms.get(code, default)]
            with pytest.raises(want) as ei:
                _lib.check(code)
            # and nothing broader on the Python side: a code Rust sends to RuntimeError is not an OSError / ValueError there
            if want is RuntimeError:
                assert not isinstance(ei.value, (OSError, ValueError, MemoryError)), name
    finally:
        _lib._lib = saved
    inte = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for c, exc in arms.items():
        assert exc in inte, exc
