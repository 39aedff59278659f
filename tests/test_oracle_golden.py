"""Pins the CPU oracle against every known-answer vector the reference holds for the path
(SURVEY.md 8c): tests/test_vfind.py:5-13 (toy.fq.gz -> MAGICAL,4) and src/lib.rs:339-369."""
import csv
import gzip
import json
import os

import pytest

from oracle import oracle, pyref


def _toy_reads(golden_dir):
    with gzip.open(os.path.join(golden_dir, "toy.fq.gz"), "rb") as f:
        lines = f.read().split(b"\n")
    return [lines[i + 1] for i in range(0, len(lines) - 1, 4) if lines[i].startswith(b"@")]


def test_reference_integration_fixture(golden_dir):
    # the reference's own test: find_variants(toy.fq.gz, adapters) == ground_truth.csv
    with open(os.path.join(golden_dir, "ground_truth.csv")) as f:
        rows = list(csv.reader(f))
    assert rows[0] == ["sequence", "count"]
    truth = {r[0].encode(): int(r[1]) for r in rows[1:]}
    got = oracle.count_fastq(os.path.join(golden_dir, "toy.fq.gz"), ("GGGCCCAGCCGGCCGGAT", "CCGGAGGCGGAGGTTCAG"))
    assert got == truth == {b"MAGICAL": 4}


def test_reference_unit_pairs(golden_dir):
    # the reference's four Rust unit tests: score vs 0.75*3*18 = 40.5
    with open(os.path.join(golden_dir, "unit_pairs.json")) as f:
        g = json.load(f)
    p = g["params"]
    for pair in g["pairs"]:
        q, t = pair["query"], pair["template"]
        score, length, eq, ed = oracle.sg_stats(q, t, p["match"], p["mismatch"], p["gap_open"], p["gap_extend"])
        min_score = p["accept"] * p["match"] * len(q)
        assert (score > min_score) == pair["pass"], pair["name"]
        assert (min_score > score) == (not pair["pass"]), pair["name"]
        d = pair["derived"]
        assert score == d["score"]
        if "length" in d:
            assert (length, ed) == (d["length"], d["end_ref"])
        # tie-free, so the value does not depend on tie policy P
        assert not pyref.tie_sensitive(q.encode(), t.encode())
        assert pyref.sg_stats_py(q.encode(), t.encode()) == (score, length, eq, ed)


def test_toy_per_read_and_derived_modes(golden_dir):
    with open(os.path.join(golden_dir, "toy_expect.json")) as f:
        g = json.load(f)
    reads = _toy_reads(golden_dir)
    assert [len(r) for r in reads] == [r["len"] for r in g["reads"]]
    st, en = oracle.spans(reads, g["adapters"])
    assert list(st) == [r["start"] for r in g["reads"]]
    assert list(en) == [r["end"] for r in g["reads"]]
    enc = lambda d: {k.encode(): v for k, v in d.items()}
    assert oracle.count(reads, g["adapters"]) == enc(g["default"])
    assert oracle.count(reads, g["adapters"], accept_prefix_alignment=1.0, accept_suffix_alignment=1.0) == enc(g["exact_only"])
    assert oracle.count(reads, g["adapters"], skip_translation=True) == enc(g["skip_translation"])
    # all DP calls on the fixture are tie-free
    for r in reads:
        for a in g["adapters"]:
            if oracle.find_exact(r, a) < 0:
                assert not pyref.tie_sensitive(a.encode(), r)


def test_translate_table():
    # spot checks from the reference table src/lib.rs:52-77 and index map :86-95
    assert oracle.translate("ATGGCGGGCATCTGTGCACTT") == "MAGICAL"
    assert oracle.translate("TAATAGTGA") == "***"
    assert oracle.translate("ATGNNNGCG") == "MXA"
    assert oracle.translate("augGCu") == "MA"
    assert oracle.translate("ATGG") is None
    assert oracle.translate("") == ""
    assert oracle.translate(b"A\xc3\xa9GGG") == "XG"
    std = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF"
    bases = "ACGT"
    for a in range(4):
        for b in range(4):
            for c in range(4):
                assert oracle.translate(bases[a] + bases[b] + bases[c]) == std[a * 16 + b * 4 + c]


def test_threshold_contract():
    # lib.rs:100-110: (0,1) -> align, ==1 -> skip, else ValueError; :260-261 f64 product order
    assert oracle.threshold(0.75, 3, 18) == (False, 41)
    assert oracle.threshold(1.0, 3, 18) == (True, 55)
    assert oracle.threshold(0.7, 3, 20) == (False, 42)   # 0.7*3.0*20.0 = 41.99999999999999
    for bad in (0.0, -0.1, 1.0000001, 2.0, float("nan"), float("inf")):
        with pytest.raises(ValueError, match="Accept alignment threshold must be between 0 and 1."):
            oracle.threshold(bad, 3, 18)


def test_exact_semantics():
    assert oracle.find_exact("ACGTACGT", "GT") == 2      # leftmost
    assert oracle.find_exact("ACGTACGT", "gt") == -1     # case-sensitive
    assert oracle.find_exact("ACG", "ACGT") == -1
    assert oracle.find_exact("ACG", "") == 0             # empty needle matches at 0
    assert oracle.find_exact("", "A") == -1


def test_reference_shape_count_equals_the_plain_count():
    """oracle.count_reference_shape (W workers -> one serial reducer, lib.rs:271-306) gives the table oracle.count gives:
    only the threading differs, and the bench times it as the CPU baseline in BASELINE.md 3's shape"""
    from tests import synth
    ads = ("GATCGGAAGAGCACACGTCT", "AGATCGGAAGAGCGTCGTGT")
    reads = synth.make_reads(5, 9000, ads, L=150, sub=0.02, indel=0.004, dirty=0.02, n_variants=300)
    want = oracle.count(reads, ads)
    for w in (1, 3, 7):
        assert oracle.count_reference_shape(reads, ads, n_workers=w) == want
    assert oracle.count_reference_shape(reads, ads, n_workers=3, skip_translation=True) == oracle.count(reads, ads, skip_translation=True)
    assert oracle.count_reference_shape([], ads) == {}
