"""Parity of the HIP path (through the C ABI) with the CPU oracle: bit-exact per-read cut
coordinates and exact (sequence,count) multisets, on the reference's fixture, on seeded synthetic
reads, and on the edge cases the domain has (short/empty/long reads, N and lowercase bytes,
adapters the 2-bit path cannot hold, exact-only mode, multi-member gzip)."""
import csv
import os

import numpy as np
import pytest

from oracle import oracle, pyref
from tests import synth

pytestmark = pytest.mark.gpu

TOY = ("GGGCCCAGCCGGCCGGAT", "CCGGAGGCGGAGGTTCAG")
AD20 = ("GATCGGAAGAGCACACGTCT", "AGATCGGAAGAGCGTCGTGT")


def _engine(adapters, **kw):
    from vfind_amd import Engine
    return Engine(adapters, table_slots=1 << 16, arena_bytes=1 << 22, **kw)


def _check_spans(reads, adapters, **kw):
    with _engine(adapters, **kw) as e:
        st, en = e.spans(reads)
    okw = {k: v for k, v in kw.items() if k not in ("batch_reads", "queue_len")}
    ost, oen = oracle.spans(reads, adapters, **okw)
    bad = np.nonzero((st != ost) | (en != oen))[0]
    assert bad.size == 0, f"first mismatch read {bad[0]}: gpu=({st[bad[0]]},{en[bad[0]]}) oracle=({ost[bad[0]]},{oen[bad[0]]}) {reads[bad[0]]!r}"
    return st, en


def _check_counts(reads, adapters, **kw):
    with _engine(adapters, **kw) as e:
        e.submit(reads)
        got = e.finish_dict()
        stats = e.stats()
    okw = {k: v for k, v in kw.items() if k not in ("batch_reads", "queue_len")}
    want = oracle.count(reads, adapters, **okw)
    assert got == want
    # one entry per key (lib.rs:296, 301): the result build sums equal strings, so a key that got into the table twice would
    # still give the right frame - it must not happen at all (see _dup_bound for the runs where a bound is all there is)
    assert stats["dup_rows"] == 0, f"{stats['dup_rows']} duplicate table entries were folded by the result build"
    # (keys that are not valid UTF-8 are counted on the device and dropped when the frame is built)
    assert stats["counted"] >= sum(want.values())
    if all(max(r, default=0) < 0x80 for r in map(bytes, reads)):
        assert stats["counted"] == sum(want.values())
    return got


def test_reference_fixture_end_to_end(golden_dir):
    import vfind_amd
    out = vfind_amd.find_variants(os.path.join(golden_dir, "toy.fq.gz"), TOY, show_progress=False)
    with open(os.path.join(golden_dir, "ground_truth.csv")) as f:
        rows = list(csv.reader(f))[1:]
    cols = {c: (out[c].to_list() if hasattr(out[c], "to_list") else out[c].to_pylist()) for c in ("sequence", "count")}
    assert list(zip(cols["sequence"], cols["count"])) == [(r[0], int(r[1])) for r in rows] == [("MAGICAL", 4)]
    names = out.column_names if hasattr(out, "column_names") else out.columns
    assert list(names) == ["sequence", "count"]


def test_reference_fixture_modes(golden_dir):
    import gzip
    import json
    import vfind_amd
    path = os.path.join(golden_dir, "toy.fq.gz")
    g = json.load(open(os.path.join(golden_dir, "toy_expect.json")))
    lines = gzip.open(path, "rb").read().split(b"\n")
    reads = [lines[i + 1] for i in range(0, len(lines) - 1, 4)]
    st, en = _check_spans(reads, TOY)
    assert list(st) == [r["start"] for r in g["reads"]] and list(en) == [r["end"] for r in g["reads"]]
    with vfind_amd.Engine(TOY, accept_prefix_alignment=1.0, accept_suffix_alignment=1.0) as e:
        e.run_fastq(path)
        assert e.finish_dict() == {b"MAGICAL": 1}
    with vfind_amd.Engine(TOY, skip_translation=True) as e:
        e.run_fastq(path)
        assert e.finish_dict() == {k.encode(): v for k, v in g["skip_translation"].items()}


@pytest.mark.parametrize("seed,L,sub,indel", [(1, 150, 0.005, 0.0005), (2, 150, 0.15, 0.0), (3, 150, 0.05, 0.03), (4, 300, 0.03, 0.01), (5, 600, 0.03, 0.01)])
def test_spans_parity_random(seed, L, sub, indel):
    reads = synth.make_reads(seed, 3000, AD20, L=L, sub=sub, indel=indel)
    st, en = _check_spans(reads, AD20)
    assert (st >= 0).sum() > 100 and (en >= 0).sum() > 100


@pytest.mark.parametrize("m", [8, 9, 12, 16, 17, 20, 24, 32])
def test_exact_scan_decoys(m):
    """the exact search checks (up to) eight pattern bases at every position and the whole pattern only at
    the survivors: decoys that agree with the adapter on exactly those bases - before, after, overlapping
    and at the very end of the read - must not move memmem's leftmost answer"""
    rng = np.random.default_rng(1000 + m)
    B = b"ACGT"
    def rand(k):
        return bytes(B[i] for i in rng.integers(0, 4, k))
    pre, suf = rand(m), rand(m)
    while suf == pre:
        suf = rand(m)
    # bases stage 1 looks at (vf_k_front.hip scan_plan): pairs (b, b+16) while the adapter is longer than 16 + b, then singles
    pairs = max(0, min(4, m - 16))
    singles = max(0, min(8 - 2 * pairs, min(m, 16) - pairs))
    checked = set(range(pairs + singles)) | {b + 16 for b in range(pairs)}
    def decoy(ad):
        d = bytearray(ad)
        free = [i for i in range(m) if i not in checked]
        if not free:                                   # m <= 8: stage 1 is already exact
            d[rng.integers(0, m)] = B[(B.index(bytes([d[0]])) + 1) % 4]
            return bytes(d)
        for i in rng.choice(free, size=max(1, len(free) // 2), replace=False):
            d[i] = B[(B.index(bytes([d[i]])) + 1 + rng.integers(0, 3)) % 4]
        return bytes(d)
    reads = []
    for k in range(400):
        parts = [rand(int(rng.integers(0, 12)))]
        for ad in (pre, suf):
            shape = k % 8
            if shape in (0, 1): parts += [decoy(ad), rand(int(rng.integers(0, 9))), ad]              # decoy first
            elif shape == 2: parts += [ad, rand(3), decoy(ad)]                                    # decoy after
            elif shape == 3: parts += [decoy(ad)[: m // 2], ad]                                   # half a decoy runs into the real one
            elif shape == 4: parts += [decoy(ad), decoy(ad), ad, ad]                              # two decoys, two copies
            elif shape == 5: parts += [decoy(ad)]                                                 # decoy only
            elif shape == 6: parts += [ad[:-1]]                                                   # truncated copy
            else: parts += [rand(5), ad]
            parts.append(rand(int(rng.integers(6, 30))))
        if k % 16 == 7:
            parts.append(pre)                                                                     # a copy flush with the read end
        if k % 16 == 15:
            parts.append(decoy(suf))                                                              # a decoy flush with the read end
        reads.append(b"".join(parts)[:600])
    # exact-only mode isolates the scan; the default mode adds the alignment of the decoys
    _check_spans(reads, (pre.decode(), suf.decode()), accept_prefix_alignment=1.0, accept_suffix_alignment=1.0)
    _check_spans(reads, (pre.decode(), suf.decode()))
    _check_counts(reads, (pre.decode(), suf.decode()), skip_translation=True)


def test_spans_parity_tie_sensitive_reads_follow_policy_P():
    # reads whose optimal alignment is not unique: the kernel must agree with the oracle's tie
    # policy P, and the length it reports must be one of the co-optimal lengths
    reads = synth.make_reads(11, 1500, AD20, L=100, sub=0.06, indel=0.06, flank_max=7)
    st, en = _check_spans(reads, AD20)
    n_tie = 0
    for i, r in enumerate(reads[:400]):
        if oracle.find_exact(r, AD20[0]) >= 0 or st[i] < 0:
            continue
        score, ends = pyref.cooptimal(AD20[0].encode(), r)
        lens = {e[2] for e in ends}
        assert int(st[i]) in lens
        n_tie += len(lens) > 1
    assert n_tie > 0, "test input holds no tie-sensitive read"


def test_counts_parity_translate_and_skip():
    reads = synth.make_reads(21, 20000, AD20, L=150, sub=0.01, indel=0.002, n_variants=300)
    got = _check_counts(reads, AD20)
    assert len(got) > 50
    got = _check_counts(reads, AD20, skip_translation=True)
    assert len(got) > 50


def test_counts_parity_variable_length_and_dirty_reads():
    reads = synth.make_reads(22, 8000, AD20, L=150, sub=0.02, indel=0.005, vary_len=True, dirty=0.2)
    _check_spans(reads, AD20)
    _check_counts(reads, AD20)
    _check_counts(reads, AD20, skip_translation=True)


def test_exact_only_six_bp_adapters_cfg1():
    # BASELINE.json configs[0]: 10k reads, 150 bp, 6 bp adapters, accept=1.0 (exact only)
    ads = ("GGATCC", "CTCGAG")
    reads = synth.make_reads(31, 10000, ads, L=150, sub=0.0, indel=0.0, n_variants=1000)
    kw = dict(accept_prefix_alignment=1.0, accept_suffix_alignment=1.0)
    _check_spans(reads, ads, **kw)
    got = _check_counts(reads, ads, **kw)
    assert len(got) > 100


def test_edge_reads():
    reads = [b"", b"A", b"ACGT", AD20[0].encode(), AD20[1].encode(), (AD20[0] + AD20[1]).encode(),
             (AD20[0] + "ATG" + AD20[1]).encode(), (AD20[0] + "ATGA" + AD20[1]).encode(),
             (AD20[1] + "ATGAAA" + AD20[0]).encode(),                       # suffix before prefix
             (AD20[0] + "ATGAAA" + AD20[1] + "CC" + AD20[1]).encode(),      # leftmost suffix wins
             b"N" * 150, b"acgt" * 30, (AD20[0][:19] + "N" + "ATGAAA" + AD20[1]).encode(),
             (AD20[0] + "ATGNNNAAA" + AD20[1]).encode(), (AD20[0] + "atgaaa" + AD20[1]).encode(),
             (AD20[0] + "AUGAAA" + AD20[1]).encode(), (AD20[0].lower() + "ATGAAA" + AD20[1]).encode(),
             (AD20[0] + "ATG" * 250 + AD20[1]).encode(),                    # 790 bp: byte-wise path
             (AD20[0] + "ATG" * 60 + AD20[1]).encode(),                     # 220 bp: W=20
             (AD20[0] + "\xc3\xa9ATGA" + AD20[1]).encode("latin-1"),         # non-ASCII bytes
             (AD20[0] + "\xff\xfeAAAA" + AD20[1]).encode("latin-1")]         # invalid UTF-8 region
    _check_spans(reads, AD20)
    _check_counts(reads, AD20)
    _check_counts(reads, AD20, skip_translation=True)
    with _engine(AD20) as e:
        e.submit([])
        assert e.finish_dict() == {}


@pytest.mark.parametrize("ads", [("ACGTNACGTACGTACGTACG", "TTGACCAGTNNACGATTGCA"), ("A" * 33 + "CGT", "TGCA" * 10), ("acgtacgtacgtacgt", "TGCATGCATGCATGCA"), ("", "ACGT"), ("ACGTACGTAC", "")])
def test_adapters_outside_the_fast_path(ads):
    rd = synth.make_reads(41, 600, ("ACGTTACGTACGTACGTACG", "TTGACCAGTCAACGATTGCA"), L=120)
    rd += synth.make_reads(42, 400, [a or "ACGT" for a in ads], L=120, sub=0.02)
    _check_spans(rd, ads)
    _check_counts(rd, ads)


@pytest.mark.parametrize("kw", [dict(match_score=1, mismatch_score=-1, gap_open_penalty=2, gap_extend_penalty=1),
                                dict(match_score=5, mismatch_score=-4, gap_open_penalty=10, gap_extend_penalty=1, accept_prefix_alignment=0.6, accept_suffix_alignment=0.9),
                                dict(match_score=2, mismatch_score=-3, gap_open_penalty=0, gap_extend_penalty=0),
                                dict(match_score=300, mismatch_score=-200, gap_open_penalty=500, gap_extend_penalty=200),
                                dict(accept_prefix_alignment=0.7, accept_suffix_alignment=1.0)])
def test_scoring_parameters(kw):
    reads = synth.make_reads(51, 2500, AD20, L=150, sub=0.06, indel=0.02)
    _check_spans(reads, AD20, **kw)
    _check_counts(reads, AD20, **kw)


@pytest.mark.parametrize("m", [1, 2, 7, 16, 17, 31, 32])
def test_adapter_lengths(m):
    rng = np.random.default_rng(m)
    ads = (synth.rand_seq(rng, m).decode(), synth.rand_seq(rng, m).decode())
    reads = synth.make_reads(60 + m, 2000, ads, L=120, sub=0.04, indel=0.01)
    _check_spans(reads, ads, accept_prefix_alignment=0.8, accept_suffix_alignment=0.8)


def test_filter_and_window_edge_cases():
    """Myers pre-filter + windowed DP must equal the full DP: adapters overhanging either read end,
    long insertions, several mutated copies far apart (hull wider than the window -> full-width DP),
    windows touching column 0 and the last column"""
    rng = np.random.default_rng(5)
    P, S = AD20
    mut = lambda a, i: a[:i] + ("A" if a[i] != "A" else "C") + a[i + 1:]
    R = lambda k: synth.rand_seq(rng, k).decode()
    reads = [mut(P, 3) + R(60) + mut(S, 10) + R(30),                 # window starts at column 0
             P[3:] + R(70) + mut(S, 2) + R(20),                      # prefix overhangs the read start
             R(5) + mut(P, 7) + R(60) + S[:17],                      # suffix overhangs the read end
             R(9) + mut(P, 1) + R(40) + mut(P, 15) + R(30) + mut(S, 4) + R(4) + mut(S, 9),   # copies far apart
             R(8) + P[:10] + "ACGTA" + P[10:] + R(50) + mut(S, 5) + R(20),                   # 5-base insertion
             R(8) + P[:9] + P[11:] + R(50) + S[:12] + S[13:] + R(30),                        # deletions
             mut(mut(P, 2), 12) + R(100) + mut(mut(S, 5), 17),       # two mismatches each, at both read ends
             R(130) + mut(P, 5),                                     # prefix-like match at the very end
             mut(S, 5) + R(130),                                     # suffix-like match at the very start
             P[:15], S[5:], R(18), mut(P, 0), mut(S, 19)]
    reads = [r.encode() for r in reads]
    for kw in (dict(), dict(accept_prefix_alignment=0.6, accept_suffix_alignment=0.6),
               dict(accept_prefix_alignment=0.9, accept_suffix_alignment=0.5),
               dict(match_score=2, mismatch_score=-1, gap_open_penalty=3, gap_extend_penalty=1)):
        _check_spans(reads, AD20, **kw)
        _check_counts(reads, AD20, **kw)
    # randomised: heavy indel noise and flanks on both sides, several thresholds
    for seed, acc in ((201, 0.5), (202, 0.65), (203, 0.75), (204, 0.9)):
        rd = synth.make_reads(seed, 3000, AD20, L=150, sub=0.08, indel=0.08, flank_max=40, vary_len=True)
        _check_spans(rd, AD20, accept_prefix_alignment=acc, accept_suffix_alignment=acc)


def test_banded_class_single_edit_tasks():
    """Tasks whose best end column has unit edit distance 1 take the banded DP (dp_banded in k_dpall, class 0 of the filter):
    one substitution, one inserted or one deleted base at every adapter position, with the adapter at the read
    start (band left of column 0), at the read end (last-column cells, every read length mod 4), overhanging
    either end, and next to a second, worse copy; several scoring schemes and adapter lengths"""
    rng = np.random.default_rng(77)
    R = lambda k: synth.rand_seq(rng, k).decode()
    other = lambda c: "ACGT"[("ACGT".index(c) + 1 + int(rng.integers(0, 3))) % 4]
    for ads in (AD20, ("GATCGGAAGAGCACACGTCTGAACTCCAGT", "AGATCGGAAGAGCGTCGTGTAGGGAAAGAG"), ("ACGGTCA", "TTGCAAGC")):
        P, S = ads
        reads = []
        for ad, is_p in ((P, True), (S, False)):
            m = len(ad)
            oth = S if is_p else P
            for i in range(m):
                for kind in range(3):
                    e = ad[:i] + other(ad[i]) + ad[i + 1:] if kind == 0 else (ad[:i] + R(1) + ad[i:] if kind == 1 else ad[:i] + ad[i + 1:])
                    for flank in (0, 1, 2, 3, 7):
                        for tail in (0, 1, 2, 3, 5):
                            body = R(flank) + e + R(tail) if not is_p else R(flank) + e + R(30 + tail) + oth
                            if not is_p:
                                body = oth + R(33) + body
                            reads.append(body)
                    reads.append(e[2:] + R(40) + oth if is_p else oth + R(40) + e[:-2])          # overhang + one edit
            reads.append(R(3) + ad[:m // 2] + other(ad[m // 2]) + ad[m // 2 + 1:] + R(m + 9) + ad[:3] + other(ad[3]) + other(ad[4]) + ad[5:] + R(20))
        reads = [r.encode() for r in reads]
        for kw in (dict(), dict(accept_prefix_alignment=0.6, accept_suffix_alignment=0.9),
                   dict(match_score=2, mismatch_score=-1, gap_open_penalty=3, gap_extend_penalty=1),
                   dict(match_score=5, mismatch_score=-1, gap_open_penalty=5, gap_extend_penalty=2, accept_prefix_alignment=0.6, accept_suffix_alignment=0.6),
                   dict(match_score=4, mismatch_score=-4, gap_open_penalty=8, gap_extend_penalty=4, accept_prefix_alignment=0.5, accept_suffix_alignment=0.5)):
            _check_spans(reads, ads, **kw)
    # random low-error libraries: most tasks are class 0; read lengths of every residue mod 16
    for seed, L in ((301, 150), (302, 151), (303, 157), (304, 160), (305, 98)):
        rd = synth.make_reads(seed, 4000, AD20, L=L, sub=0.02, indel=0.01, flank_max=9, vary_len=True)
        _check_spans(rd, AD20)
        _check_counts(rd, AD20)


def test_hamming_certified_band_two_substitution_tasks():
    """Round 4: a single-cluster task whose cluster's centre diagonal carries the adapter with h >= 2 substitutions takes a band of
    (hi - lo) + 2 kq + 1 diagonals instead of a 64-column window (hamming_band_class in the Myers pass).  EVERY
    pair of substituted adapter positions, for both adapters, with the adapter at the read start (band left of column 0), at
    the read end (last-column cells), behind short and long flanks, overhanging either end, next to a worse second copy;
    three and four substitutions, a substitution pair next to an indel (no certificate: the window classes), several scoring
    schemes - spans against the oracle's full DP"""
    rng = np.random.default_rng(20261005)
    R = lambda k: synth.rand_seq(rng, k).decode()
    other = lambda c: "ACGT"[("ACGT".index(c) + 1 + int(rng.integers(0, 3))) % 4]

    def subs(ad, pos):
        a = list(ad)
        for p in pos:
            a[p] = other(a[p])
        return "".join(a)
    for ads in (AD20, ("GATCGGAAGAGCACACGTCTGAACTCCAGT", "AGATCGGAAGAGCGTCGTGTAGGGAAAGAG"), ("ACGGTCATGCA", "TTGCAAGCGTAC")):
        P, S = ads
        reads = []
        for ad, is_p in ((P, True), (S, False)):
            m = len(ad)
            oth = S if is_p else P
            k = 0
            for i in range(m):
                for j in range(i + 1, m):
                    e = subs(ad, (i, j))
                    flank, tail = (0, 1, 2, 5, 11, 40)[k % 6], (0, 1, 3, 6, 2, 0)[(k // 6) % 6]
                    k += 1
                    body = R(flank) + e + R(30 + tail) + oth if is_p else oth + R(33) + R(flank) + e + R(tail)
                    reads.append(body)
            for t in range(60):                                                                   # three and four substitutions, cut copies
                pos = sorted(rng.choice(m, size=3 + t % 2, replace=False).tolist())
                e = subs(ad, pos)
                reads.append(R(t % 7) + e + R(35) + oth if is_p else oth + R(35) + e + R(t % 5))
                e2 = subs(ad, pos[:2])
                reads.append(e2[1 + t % 3:] + R(40) + oth if is_p else oth + R(40) + e2[: m - 1 - t % 3])      # two substitutions + overhang
                q = int(rng.integers(1, m - 1))
                e3 = e2[:q] + (R(1) if t % 2 else "") + e2[q + (0 if t % 2 else 1):]               # + one inserted / deleted base
                reads.append(R(3) + e3 + R(35) + oth if is_p else oth + R(35) + e3 + R(2))
                reads.append(R(2) + e2 + R(t % 4) + subs(ad, pos) + R(30) + oth if is_p else oth + R(30) + subs(ad, pos) + R(t % 4) + e2)   # second copy close by
        reads = [r.encode() for r in reads]
        for kw in (dict(), dict(accept_prefix_alignment=0.6, accept_suffix_alignment=0.7),
                   dict(match_score=2, mismatch_score=-1, gap_open_penalty=3, gap_extend_penalty=1, accept_prefix_alignment=0.6, accept_suffix_alignment=0.6),
                   dict(match_score=5, mismatch_score=-1, gap_open_penalty=5, gap_extend_penalty=2, accept_prefix_alignment=0.6, accept_suffix_alignment=0.6),
                   dict(match_score=4, mismatch_score=-4, gap_open_penalty=8, gap_extend_penalty=4, accept_prefix_alignment=0.5, accept_suffix_alignment=0.5)):
            _check_spans(reads, ads, **kw)
        _check_counts(reads, ads)
    # a library with 15 % substitutions (the cfg-4 shape): most accepted tasks carry two substitutions; every read length mod 16
    for seed, L in ((401, 150), (402, 151), (403, 157), (404, 160), (405, 98)):
        rd = synth.make_reads(seed, 4000, AD20, L=L, sub=0.15, indel=0.0, flank_max=9, vary_len=True)
        _check_spans(rd, AD20)
        _check_counts(rd, AD20)


def test_indel_scan_at_its_threshold():
    """Round 4: the pre-DP scan measures indel distance where the scoring scheme allows (a mismatch = 2 units; myers_scan<W, true>,
    DESIGN.md 4.3 "the indel scan").  The alignments that DEFINE its threshold - at the defaults five units: a five-base insertion,
    a substitution and a three-base insertion, two two-base insertions, three unaligned end bases and a substitution, one and two
    substitutions - and the ones just beyond it (a six-base insertion, three substitutions, a substitution and a four-base
    insertion), at every position of both adapters, at the read start and end, behind flanks of every length mod 16; schemes for
    which the host keeps the unit scan (cheap substitutions) and schemes that change the threshold; spans against the oracle's
    full DP.  Interior three-substitution copies - what the scan is there to turn away - are rejected, by it or by the DP"""
    rng = np.random.default_rng(20261009)
    R = lambda k: synth.rand_seq(rng, k).decode()
    other = lambda c: "ACGT"[("ACGT".index(c) + 1 + int(rng.integers(0, 3))) % 4]

    def sub(a, p):
        return a[:p] + other(a[p]) + a[p + 1:]
    for ads in (AD20, ("GATCGGAAGAGCACACGTCTGAACTCCAGT", "AGATCGGAAGAGCGTCGTGTAGGGAAAGAG"), ("ACGGTCATGCATT", "TTGCAAGCGTACG")):
        P, S = ads
        reads = []
        for ad, is_p in ((P, True), (S, False)):
            m = len(ad)
            oth = S if is_p else P
            k = 0

            def put(e):
                nonlocal k
                flank, tail = k % 17, (0, 1, 3, 6, 2, 0, 9)[(k // 17) % 7]
                k += 1
                reads.append(R(flank) + e + R(30 + tail) + oth if is_p else oth + R(33) + R(flank) + e + R(tail))
            for q in range(1, m):
                for g in (3, 4, 5, 6):
                    put(ad[:q] + R(g) + ad[q:])                                   # one insertion run
                    p = int(rng.integers(0, m))
                    put(sub(ad, p)[:q] + R(g - 2 if g > 3 else g) + sub(ad, p)[q:])   # a substitution and a run
                q2 = int(rng.integers(1, m))
                a, b = sorted((q, q2))
                put(ad[:a] + R(2) + ad[a:b] + R(2) + ad[b:])                       # two runs
                put(ad[:a] + R(1) + ad[a:b] + R(3) + ad[b:])
                put(ad[:q] + ad[q + 1:])                                          # deletions
                put(sub(ad, (q + 3) % m)[:q] + sub(ad, (q + 3) % m)[q + 1:])
                if q + 2 <= m:
                    put(ad[:q] + ad[q + 2:])
            for u in (1, 2, 3, 4, 5):                                             # unaligned end bases (+ substitutions)
                for t in range(6):
                    e = ad
                    for _ in range(t % 3):
                        e = sub(e, int(rng.integers(u, m - u)))
                    reads.append(e[u:] + R(40) + oth if is_p else oth + R(40) + e[: m - u])
            for t in range(80):                                                    # three substitutions away from the ends
                pos = sorted(rng.choice(np.arange(2, m - 2), size=3, replace=False).tolist())
                e = ad
                for p in pos:
                    e = sub(e, p)
                put(e)
        reads = [r.encode() for r in reads]
        for kw in (dict(), dict(accept_prefix_alignment=0.6, accept_suffix_alignment=0.85),
                   dict(match_score=2, mismatch_score=-1, gap_open_penalty=3, gap_extend_penalty=1, accept_prefix_alignment=0.6, accept_suffix_alignment=0.6),
                   dict(match_score=1, mismatch_score=0, gap_open_penalty=4, gap_extend_penalty=2, accept_prefix_alignment=0.7, accept_suffix_alignment=0.7),
                   dict(match_score=5, mismatch_score=-4, gap_open_penalty=2, gap_extend_penalty=1, accept_prefix_alignment=0.7, accept_suffix_alignment=0.6),
                   dict(match_score=4, mismatch_score=-4, gap_open_penalty=8, gap_extend_penalty=4, accept_prefix_alignment=0.5, accept_suffix_alignment=0.5)):
            _check_spans(reads, ads, **kw)
        _check_counts(reads, ads)
    # three substitutions in the adapter's interior, flanks on both sides: the scheme rejects every one of them (loss 15 > 14); the
    # unit-distance scan sent every one to the DP (distance 3 <= 5), the indel scan those whose rows happen to line up one column off
    # (59 % in the CPU model; the class histogram of profiles/r04_cfg4_kernel_trace.md shows what is left)
    P, S = AD20
    reads = []
    for t in range(4000):
        pos = sorted(rng.choice(np.arange(2, 18), size=3, replace=False).tolist())
        e = P
        for p in pos:
            e = sub(e, p)
        reads.append((R(12 + t % 9) + e + R(60) + S + R(10)).encode())
    st, en = _check_spans(reads, AD20)
    assert (st < 0).mean() > 0.99              # (next to none of them has a prefix: rejected by the scan or by the DP, as the oracle's full DP does)


@pytest.mark.parametrize("name,kw", [
    ("cfg2", dict(seed=0x5EED0002, read_len=150, m=20, sub_rate=0.005, indel_rate=0.0005)),
    ("cfg4", dict(seed=0x5EED0004, read_len=150, m=20, sub_rate=0.15, indel_rate=0.0)),
    ("cfg5", dict(seed=0x5EED0005, read_len=300, m=30, sub_rate=0.005, indel_rate=0.0005))])
def test_baseline_config_shapes_small_scale(name, kw):
    """BASELINE.json configs[1], [3], [4] at 40 k reads through the bench's own generator: spans and the
    (sequence,count) table equal the oracle; full sizes are covered by bench.py's parity gate"""
    from benchlib import synth as vsynth
    ads = vsynth.adapters(kw["seed"], kw["m"])
    gen = vsynth.Generator(kw["seed"], ads, read_len=kw["read_len"], n_variants=5000, sub_rate=kw["sub_rate"], indel_rate=kw["indel_rate"])
    data, offs = gen.ascii(0, 40000, 8)
    with _engine(ads) as e:
        st, en = e.spans((data, offs))
        e.submit((data, offs))
        got = e.finish_dict()
        stats = e.stats()
    ost, oen = oracle.spans((data, offs), ads, n_threads=8)
    assert (st == ost).all() and (en == oen).all()
    want = oracle.count((data, offs), ads, n_threads=8)
    assert got == want and len(got) > 100
    # the packed-row generator output equals host packing of the ASCII output
    rows, lens = gen.packed(0, 2000, 4)
    with _engine(ads) as e:
        e.submit_packed(rows, lens)
        assert e.finish_dict() == oracle.count((data[:2000 * kw["read_len"]], offs[:2001]), ads)
    assert stats["dp_prefix"] > 0


def test_small_batches_and_slot_reuse():
    reads = synth.make_reads(71, 5000, AD20, L=150, sub=0.02, indel=0.005, dirty=0.05)
    _check_counts(reads, AD20, batch_reads=257, queue_len=3)
    _check_spans(reads, AD20, batch_reads=129)


def test_fastq_end_to_end_multimember_gzip(tmp_path):
    import vfind_amd
    reads = synth.make_reads(81, 30000, AD20, L=150, sub=0.01, indel=0.002, n_variants=400, dirty=0.01)
    p = str(tmp_path / "multi.fq.gz")
    synth.write_fastq_gz(p, reads, members=3)
    want = oracle.count_fastq(p, AD20)
    assert want == oracle.count(reads, AD20)
    with vfind_amd.Engine(AD20, n_threads=4) as e:
        e.run_fastq(p)
        assert e.finish_dict() == want
        assert e.stats()["reads"] == len(reads)
    p2 = str(tmp_path / "crlf.fq.gz")
    synth.write_fastq_gz(p2, reads[:500], crlf=True)
    with vfind_amd.Engine(AD20) as e:
        e.run_fastq(p2)
        assert e.finish_dict() == oracle.count(reads[:500], AD20)
    out = vfind_amd.find_variants(p, AD20, show_progress=False)
    seqs = out["sequence"].to_list() if hasattr(out["sequence"], "to_list") else out["sequence"].to_pylist()
    cnts = out["count"].to_list() if hasattr(out["count"], "to_list") else out["count"].to_pylist()
    assert [(s.encode(), c) for s, c in zip(seqs, cnts)] == list(oracle.sorted_table(want))


def test_fastq_parallel_member_decode(tmp_path):
    """the reader inflates gzip members in parallel: BGZF, many small members, members far larger than
    the task grid, gzip-magic bytes inside the data, and a single member must all give the oracle's table"""
    import gzip
    import vfind_amd
    reads = synth.make_reads(83, 120000, AD20, L=150, sub=0.01, indel=0.002, n_variants=500)
    want = oracle.count(reads, AD20)
    text = synth.fastq_text(reads)
    files = {}
    p = str(tmp_path / "bgzf.fq.gz"); synth.write_bgzf(p, text); files["bgzf"] = p
    p = str(tmp_path / "many.fq.gz"); synth.write_fastq_gz(p, reads, members=97); files["97 members"] = p
    p = str(tmp_path / "single.fq.gz"); open(p, "wb").write(gzip.compress(text, 1)); files["single"] = p
    # stored (level 0) members keep the text verbatim, so gzip magic bytes in a header line appear in
    # the compressed file as false member starts
    tricky = synth.fastq_text(reads[:60000], header_extra=b" \x1f\x8b\x08\x04 x") + synth.fastq_text(reads[60000:], first=60000)
    p = str(tmp_path / "magic.fq.gz")
    open(p, "wb").write(gzip.compress(tricky[: len(tricky) // 2], 0) + gzip.compress(tricky[len(tricky) // 2:], 0))
    files["false magic"] = p
    for name, path in files.items():
        assert oracle.count_fastq(path, AD20) == want, name
        for nt in (1, 4, 16):
            with vfind_amd.Engine(AD20, n_threads=nt, table_slots=1 << 16) as e:
                e.run_fastq(path)
                assert e.finish_dict() == want, (name, nt)
                assert e.stats()["reads"] == len(reads)
    # trailing garbage after the last member is an error, as for the reference's MultiGzDecoder
    p = str(tmp_path / "garbage.fq.gz")
    open(p, "wb").write(gzip.compress(text[:100000]) + b"this is not gzip")
    with pytest.raises(vfind_amd.VfindError, match="gzip"):
        vfind_amd.find_variants(p, AD20)


def test_fastq_single_member_decoded_by_all_threads(tmp_path, monkeypatch):
    """an ordinary single-member .fastq.gz through the big-member regime (every member decoded by all threads,
    vf_pinflate.cpp), forced on and with small cells so that this 40 MB text spans many of them"""
    import gzip
    import vfind_amd
    monkeypatch.setenv("VFIND_INFLATE_PARALLEL", "1")
    monkeypatch.setenv("VFIND_PINFLATE_CELL", "262144")
    reads = synth.make_reads(84, 120000, AD20, L=150, sub=0.01, indel=0.002, n_variants=800)
    want = oracle.count(reads, AD20)
    text = synth.fastq_text(reads)
    for level in (1, 6):
        p = str(tmp_path / f"single{level}.fq.gz")
        open(p, "wb").write(gzip.compress(text, level))
        for nt in (3, 16):
            with vfind_amd.Engine(AD20, n_threads=nt, table_slots=1 << 16) as e:
                e.run_fastq(p)
                assert e.finish_dict() == want, (level, nt)
                assert e.stats()["reads"] == len(reads)
    p = str(tmp_path / "cut.fq.gz")
    open(p, "wb").write(gzip.compress(text, 6)[:2_000_000])
    with pytest.raises(vfind_amd.VfindError, match="gzip"):
        vfind_amd.find_variants(p, AD20, n_threads=8)


def test_fastq_device_framing_edge_cases(tmp_path):
    """records are framed and packed on the device: reads from 0 to 700 bases in one file (rows of 10, 20 and 40 words,
    and longer ones that only the byte-wise kernel can hold), reads with N and lowercase letters, a last record
    without its newline, trailing blank lines, and adapters the 2-bit path cannot hold at all (every read byte-wise)"""
    import gzip
    import vfind_amd
    rng = np.random.default_rng(91)
    reads = []
    for L in (0, 1, 19, 40, 150, 151, 160, 161, 300, 320, 321, 639, 640, 641, 700):
        reads += synth.make_reads(int(rng.integers(1, 1 << 30)), 60, AD20, L=max(L, 41), sub=0.02, indel=0.01, n_variants=20, dirty=0.1)
        reads += [r[:L] for r in synth.make_reads(int(rng.integers(1, 1 << 30)), 10, AD20, L=max(L, 41), n_variants=5)]
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    text = synth.fastq_text(reads)
    shapes = {"plain": text, "no final newline": text[:-1], "trailing blank lines": text + b"\n\n\r\n"}
    want = oracle.count(reads, AD20)
    for name, t in shapes.items():
        p = str(tmp_path / (name.replace(" ", "_") + ".fq.gz"))
        open(p, "wb").write(gzip.compress(t, 6))
        with vfind_amd.Engine(AD20, n_threads=3, table_slots=1 << 14) as e:
            e.run_fastq(p)
            assert e.finish_dict() == want, name
            st = e.stats()
            assert st["reads"] == len(reads) and st["reads_generic"] > 0
    odd = ("ACGTNACGTACGTACGTACG", "TTGACCAGTNNACGATTGCA")
    p = str(tmp_path / "plain.fq.gz")
    with vfind_amd.Engine(odd, n_threads=3, table_slots=1 << 14) as e:
        e.run_fastq(p)
        assert e.finish_dict() == oracle.count(reads, odd)
    bad = str(tmp_path / "blank_inside.fq.gz")
    k = text.index(b"\n@", len(text) // 2) + 1
    open(bad, "wb").write(gzip.compress(text[:k] + b"\n" + text[k:], 6))
    with pytest.raises(vfind_amd.VfindError, match="malformed FASTQ"):
        vfind_amd.find_variants(bad, AD20)


def test_fastq_errors(tmp_path):
    import gzip
    import vfind_amd
    plain = tmp_path / "plain.fq"
    plain.write_bytes(b"@r\nACGT\n+\nFFFF\n")
    with pytest.raises(vfind_amd.VfindError, match="gzip"):
        vfind_amd.find_variants(str(plain), AD20)
    for name, body in [("noat.fq.gz", b"r\nACGT\n+\nFFFF\n"), ("qual.fq.gz", b"@r\nACGT\n+\nFFF\n"), ("trunc.fq.gz", b"@r\nACGT\n+\n"), ("plus.fq.gz", b"@r\nACGT\n-\nFFFF\n")]:
        p = tmp_path / name
        p.write_bytes(gzip.compress(body))
        with pytest.raises(vfind_amd.VfindError, match="malformed FASTQ"):
            vfind_amd.find_variants(str(p), AD20)
    good = gzip.compress(b"@r\nACGT\n+\nFFFF\n" * 2000)
    p = tmp_path / "cut.fq.gz"
    p.write_bytes(good[: len(good) // 2])
    with pytest.raises(vfind_amd.VfindError):
        vfind_amd.find_variants(str(p), AD20)
    p = tmp_path / "nonl.fq.gz"
    p.write_bytes(gzip.compress(b"@r\n" + (AD20[0] + "ATGAAA" + AD20[1]).encode() + b"\n+\n" + b"F" * 46))
    out = vfind_amd.Engine(AD20)
    out.run_fastq(str(p))
    assert out.finish_dict() == {b"MK": 1}
    out.close()


def test_table_export_merge_and_clear():
    a = synth.make_reads(91, 6000, AD20, L=150, n_variants=200)
    b = synth.make_reads(92, 6000, AD20, L=150, n_variants=200)
    import torch
    with _engine(AD20) as e1, _engine(AD20) as e2:
        e1.submit(a)
        e2.submit(b)
        n = e2.export_size()
        buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        assert e2.export_to(buf.data_ptr(), n + 64) == n
        e1.merge_from(buf.data_ptr(), n)
        assert e1.finish_dict() == oracle.count(a + b, AD20)
        e1.clear()
        assert e1.finish_dict() == {}
        e1.submit(b)
        assert e1.finish_dict() == oracle.count(b, AD20)


def test_clear_switches_to_the_spare_table():
    """vf_table_clear hands the next batches an empty second table and clears the used one beside them: round after round
    the counts are those of that round's reads alone, the table grows on whichever side it fills, the read statistics
    carry across the switch, and a clear with batches still in flight drops them"""
    from vfind_amd import Engine
    rounds = [synth.make_reads(300 + i, 9000, AD20, L=150, n_variants=3000 + 500 * i, sub=0.003, indel=0.0) for i in range(5)]
    with Engine(AD20, table_slots=1 << 10, arena_bytes=1 << 22, batch_reads=1500, queue_len=3) as e:
        counted = 0
        for i, reads in enumerate(rounds):
            if i == 3:
                e.submit(rounds[0])             # still in flight when the clear comes: never seen again
                counted += sum(oracle.count(rounds[0], AD20).values())
            e.clear()
            e.submit(reads)
            want = oracle.count(reads, AD20)
            assert e.finish_dict() == want, f"round {i}"
            counted += sum(want.values())
            assert e.stats()["counted"] == counted
        e.clear()
        assert e.finish_dict() == {}
    # resident-batch lanes (the bench's path) through the same switch: every step a cleared table, two lanes in flight
    import torch
    from benchlib import synth as vsynth
    gen = vsynth.Generator(0xBEEF, AD20, n_variants=5000)
    n = 60000
    wins = []
    for w in range(3):
        rows, lens = gen.packed(w * n, n, 4)
        data, offs = gen.ascii(w * n, n, 4)
        wins.append((torch.from_numpy(rows).cuda(), torch.from_numpy(lens.view("int16")).cuda(), rows.shape[1],
                     oracle.count((data, offs), AD20, n_threads=4)))
    with Engine(AD20, table_slots=1 << 16) as e:            # (a resident batch must fit the table it meets)
        for w in (0, 1, 2, 1, 0, 2):
            dr, dl, W, want = wins[w]
            e.clear()
            e.process_device(dr.data_ptr(), W, dl.data_ptr(), n)
        assert e.finish_dict() == wins[2][3]                # only the last step's reads are in the table
        e.clear()
        for w in (0, 1):                                    # no clear in between: both batches in one table
            dr, dl, W, want = wins[w]
            e.process_device(dr.data_ptr(), W, dl.data_ptr(), n)
        both = dict(wins[0][3])
        for k, v in wins[1][3].items():
            both[k] = both.get(k, 0) + v
        assert e.finish_dict() == both


def test_counts_are_exact_pass_after_pass_on_two_contexts_at_once():
    """the count path under load (what scripts/count_stress.py does by hand): the same resident batch counted 30 times into a
    cleared table, on two contexts driven by two threads at once - table kernels of both contexts, their count logs and their
    clears interleave on the device.  Every pass of both contexts must give the oracle's table and account for every read the
    kernels say they counted"""
    import threading
    import torch
    from vfind_amd import Engine
    from benchlib import synth as vsynth
    ads = vsynth.adapters(0x5EED0002, 20)
    gen = vsynth.Generator(0x5EED0002, ads, n_variants=30_000)
    n = 300_000
    rows_h, lens_h = gen.packed(0, n, 8)
    data, offs = gen.ascii(0, n, 8)
    want = oracle.count((data, offs), ads, n_threads=8)
    rows = torch.from_numpy(rows_h).cuda()
    lens = torch.from_numpy(lens_h.view(np.int16)).cuda()
    torch.cuda.synchronize()
    bad = []

    def run(tag):
        with Engine(ads, table_slots=1 << 17) as e:
            for it in range(30):
                e.clear()
                c0 = e.stats()["counted"]
                e.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n)
                e.sync()
                got = e.finish_dict()
                if got != want or sum(got.values()) != e.stats()["counted"] - c0:
                    bad.append((tag, it, len(got), sum(got.values())))
    th = [threading.Thread(target=run, args=(t,)) for t in ("a", "b")]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad[:5]


def test_clear_by_epoch_over_forty_rounds(monkeypatch):
    """vf_table_clear of a letter-key table moves to the next epoch instead of zeroing the slot lines (every 15th clear zeroes):
    stale lines must read as free slots and stale key dwords as "not written yet".  Forty rounds on one context - the same
    slots reused round after round by other keys (a small table, few variants per round drawn from a shared pool, so stale
    lines of the SAME key and of OTHER keys with the same home slot both occur), growth in some rounds, a merge and an export in
    between - each round's table equals the oracle's; the same with skip_translation (packed keys: cleared by zeroing)"""
    import torch
    from vfind_amd import Engine
    pools = [synth.make_reads(900 + i, 2500, AD20, L=150, n_variants=40 + 25 * (i % 7), sub=0.004, indel=0.0) for i in range(8)]
    for kw, rounds, tags in ((dict(), 40, "1"), (dict(skip_translation=True), 40, "1"), (dict(), 6, "0")):      # last: the A/B switch (zeroing everywhere)
        monkeypatch.setenv("VFIND_EPOCH_TAGS", tags)
        with Engine(AD20, table_slots=1 << 8, arena_bytes=1 << 22, batch_reads=700, queue_len=3, **kw) as e, Engine(AD20, table_slots=1 << 12, **kw) as other:
            for rnd in range(rounds):
                reads = pools[rnd % 8] + pools[(3 * rnd + 1) % 8][: 300 + 40 * (rnd % 5)]
                e.clear()
                e.submit(reads)
                want = oracle.count(reads, AD20, **kw)
                if rnd % 9 == 4:                                   # a peer's table merged into this epoch's
                    extra = pools[(rnd + 2) % 8][:500]
                    other.clear()
                    other.submit(extra)
                    n = other.export_size()
                    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
                    other.export_to(buf.data_ptr(), n + 64)
                    e.merge_from(buf.data_ptr(), n)
                    for k, v in oracle.count(extra, AD20, **kw).items():
                        want[k] = want.get(k, 0) + v
                assert e.finish_dict() == want, (kw, rnd)
                assert e.stats()["table_entries"] == len(want)


def test_long_keys_over_stale_lines_of_earlier_epochs():
    """Keys longer than the 14 inline dwords keep an arena offset in their slot line.  After a clear by epoch the line a LONG
    key claims still holds what the slot's earlier key left there - the letters of a short key, or the offset of another
    long one.  Round 3: a lane whose claim CAS lost judged the w0 the CAS returned (the winner's, READY by then) against the
    line image it had loaded before the CAS, followed those stale words as an arena offset, and the process ended with
    HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.  Key words now wait for the next load (image_older) and the offset word
    validates itself (long_word_ok):
    30 rounds on one context, batches of 300 bp reads (keys of 8-81 letters, those above 56 LONG) and of 150 bp reads (8-31
    letters, all inline) taking turns on the same table, two resident batches a round so that both batch lanes insert side
    by side; every fifth round's whole table, and each of the last three, is compared with the oracle's"""
    import torch
    from vfind_amd import Engine
    from benchlib import synth as vsynth
    ads = vsynth.adapters(0x5EED0005, 20)
    gens = [vsynth.Generator(0x5EED0005 + k, ads, read_len=L, n_variants=nv, sub_rate=0.005, indel_rate=0.0005)
            for k, (L, nv) in enumerate(((300, 3_000_000), (150, 200_000), (300, 500_000), (150, 3_000_000)))]
    n = 400_000
    packed = []
    for g in gens:
        rows, lens = g.packed(0, n, 8)
        packed.append((torch.from_numpy(rows).cuda(), torch.from_numpy(lens.view("int16")).cuda(), rows.shape[1]))
    with Engine(ads, table_slots=1 << 20, arena_bytes=1 << 28) as e:
        for rnd in range(30):
            k = rnd % 4
            d_rows, d_lens, W = packed[k]
            e.clear()
            half = n // 2                                          # two resident batches: both batch lanes insert side by side
            e.process_device(d_rows.data_ptr(), W, d_lens.data_ptr(), half)
            e.process_device(d_rows.data_ptr() + half * W * 4, W, d_lens.data_ptr() + half * 2, n - half)
            if rnd % 5 == 4 or rnd >= 27:
                data, offs = gens[k].ascii(0, n, 8)
                assert e.finish_dict() == oracle.count((data, offs), ads, n_threads=16), rnd
        e.sync()
    # the same keys through the streaming path into a table that starts at 4 096 slots: growth re-inserts LONG lines (k_rehash
    # reads the offset word, table_find_or_insert writes it again), epoch after epoch
    with Engine(ads, table_slots=1 << 12, arena_bytes=1 << 20, batch_reads=20_000, queue_len=3) as e:
        for rnd in range(6):
            g = gens[(2 * rnd) % 4]                                 # 300 bp batches only
            data, offs = g.ascii(rnd * 1000, 90_000, 8)
            e.clear()
            e.submit((data, offs))
            assert e.finish_dict() == oracle.count((data, offs), ads, n_threads=16), rnd


def test_count_log_both_modes_follow_the_oracle():
    """k_finish writes occurrences of known keys to the batch's count log; the log is applied by slot ranges (k_log_scatter +
    k_log_apply) or, for tables far larger than a batch, entry by entry (k_log_direct).  Both forced in turn (VFIND_COUNT_LOG,
    an A/B switch) over streaming submits, resident batches, a merge and table growth in between: the counts are the
    oracle's either way"""
    import subprocess
    import sys
    from tests.conftest import ROOT
    code = r'''
import numpy as np, torch, vfind_amd
from oracle import oracle
from tests import synth
from benchlib import synth as vsynth
AD = ("ACGTTGCAAGCTTGCATGCA", "TTGACCGTAAGCTAGGCTAA")
reads = synth.make_reads(77, 60000, AD, L=150, n_variants=4000, sub=0.003, indel=0.0)
want = oracle.count(reads, AD)
with vfind_amd.Engine(AD, table_slots=1 << 10, arena_bytes=1 << 22, batch_reads=4000, queue_len=3) as e, vfind_amd.Engine(AD) as e2:
    for lo in range(0, len(reads), 6000):
        e.submit(reads[lo:lo + 6000])
    assert e.finish_dict() == want
    e2.submit(reads[:30000])
    n = e2.export_size()
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    e2.export_to(buf.data_ptr(), n + 64)
    e.merge_from(buf.data_ptr(), n)
    e.submit(reads[:6000])
    both = dict(want)
    for k, v in oracle.count(reads[:30000], AD).items():
        both[k] = both.get(k, 0) + v
    for k, v in oracle.count(reads[:6000], AD).items():
        both[k] = both.get(k, 0) + v
    assert e.finish_dict() == both
gen = vsynth.Generator(0xF01D, AD, n_variants=3000)
n = 30000
with vfind_amd.Engine(AD, table_slots=1 << 15) as e:
    tot = {}
    for w in range(4):
        rows, lens = gen.packed(w * n, n, 4)
        data, offs = gen.ascii(w * n, n, 4)
        dr, dl = torch.from_numpy(rows).cuda(), torch.from_numpy(lens.view("int16")).cuda()
        e.process_device(dr.data_ptr(), rows.shape[1], dl.data_ptr(), n)
        e.sync()
        for k, v in oracle.count((data, offs), AD, n_threads=4).items():
            tot[k] = tot.get(k, 0) + v
    assert e.finish_dict() == tot
print("ok")
'''
    for mode in ("1", "2"):
        env = dict(os.environ, VFIND_COUNT_LOG=mode, PYTHONPATH=ROOT)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert out.returncode == 0, (mode, out.stderr[-3000:])
        assert "ok" in out.stdout


def test_table_grows_between_batches():
    """a table that fills up is quadrupled and re-hashed on the device between batches"""
    from vfind_amd import Engine
    reads = synth.make_reads(96, 60000, AD20, L=150, n_variants=20000, sub=0.002, indel=0.0)
    for kw in (dict(), dict(skip_translation=True)):
        with Engine(AD20, table_slots=1 << 10, arena_bytes=1 << 22, batch_reads=400, queue_len=3, **kw) as e:
            for lo in range(0, len(reads), 5000):
                e.submit(reads[lo:lo + 5000])
            got = e.finish_dict()
            assert e.stats()["table_entries"] == len(got)
        assert got == oracle.count(reads, AD20, **kw)
        assert len(got) > 5000


def test_exception_barrier_at_the_c_abi(monkeypatch):
    """a C++ exception inside an entry point - std::bad_alloc, a length_error from a vector sized by a bad value, anything thrown
    on a worker thread of the FASTQ reader - comes back as an error code and a message (round 2's r2o_edge.log: a garbage size
    into resize() inside vf_finish ended the Python process); the context is still usable afterwards"""
    import vfind_amd
    reads = synth.make_reads(61, 3000, AD20, L=150, n_variants=100)
    with _engine(AD20) as e:
        e.submit(reads)
        monkeypatch.setenv("VFIND_TEST_THROW", "finish")
        with pytest.raises(MemoryError, match="out of host memory"):
            e.finish_dict()
        monkeypatch.setenv("VFIND_TEST_THROW", "finish_length")
        with pytest.raises(vfind_amd.VfindError, match="internal error"):
            e.finish_dict()
        monkeypatch.delenv("VFIND_TEST_THROW")
        assert e.finish_dict() == oracle.count(reads, AD20)
    toy = os.path.join(os.path.dirname(__file__), "golden", "toy.fq.gz")
    toy_ads = ("GGGCCCAGCCGGCCGGAT", "CCGGAGGCGGAGGTTCAG")
    monkeypatch.setenv("VFIND_TEST_THROW", "fastq_worker")
    with pytest.raises(vfind_amd.VfindError, match="worker thread"):
        vfind_amd.find_variants(toy, toy_ads, show_progress=False)
    monkeypatch.delenv("VFIND_TEST_THROW")
    with vfind_amd.Engine(toy_ads) as e:
        e.run_fastq(toy)
        assert e.finish_dict() == {b"MAGICAL": 4}


def test_capacity_error_is_loud():
    """streaming submits reserve room before they are enqueued (the table grows first), so a tiny table
    still gives the oracle's answer; a resident batch larger than the table cannot be saved and fails loudly,
    (the key arena for long keys grows with the streaming path as well)"""
    import torch
    from vfind_amd import Engine, VfindError
    from benchlib import synth as vsynth
    reads = synth.make_reads(95, 4000, AD20, L=150, n_variants=2000)
    with Engine(AD20, skip_translation=True, table_slots=64, batch_reads=512) as e:
        e.submit(reads)
        assert e.finish_dict() == oracle.count(reads, AD20, skip_translation=True)
        assert e.stats()["table_entries"] > 64
    gen = vsynth.Generator(7, AD20, n_variants=3000)
    rows, lens = gen.packed(0, 20000, 4)
    d_rows = torch.from_numpy(rows).cuda()
    d_lens = torch.from_numpy(lens.view("int16")).cuda()
    with Engine(AD20, table_slots=64) as e:
        e.process_device(d_rows.data_ptr(), rows.shape[1], d_lens.data_ptr(), 20000)
        with pytest.raises(VfindError, match="table full"):
            e.finish_dict()
    # long keys live in the key arena, which grows the same way for streaming submits
    long_reads = synth.make_reads(96, 3000, AD20, L=600, n_variants=3000)
    with Engine(AD20, skip_translation=True, arena_bytes=1 << 12, table_slots=1 << 14) as e:
        e.submit(long_reads)
        got = e.finish_dict()
        assert e.stats()["arena_dwords"] > (1 << 10)
    assert got == oracle.count(long_reads, AD20, skip_translation=True)
    assert max(len(k) for k in got) > 224


@pytest.mark.parametrize("n_variants,zipf", [(300, 1.1), (200000, 0.0)])
def test_counts_large_batch_hot_and_flat_keys(n_variants, zipf):
    """a million reads through the resident path: a few hundred hot keys (LDS aggregation, per-XCD count
    arrays summed at export) and a flat library (nearly every read a new or rare key: insert path)"""
    import torch
    from vfind_amd import Engine
    from benchlib import synth as vsynth
    gen = vsynth.Generator(0xC0FFEE + n_variants, AD20, n_variants=n_variants, zipf_s=zipf)
    n = 1_000_000
    rows, lens = gen.packed(0, n, 8)
    data, offs = gen.ascii(0, n, 8)
    want = oracle.count((data, offs), AD20, n_threads=8)
    d_rows = torch.from_numpy(rows).cuda()
    d_lens = torch.from_numpy(lens.view("int16")).cuda()
    with Engine(AD20, table_slots=1 << 20) as e:
        for _ in range(2):                             # cold, then onto existing keys
            e.process_device(d_rows.data_ptr(), rows.shape[1], d_lens.data_ptr(), n)
        got = e.finish_dict()
        st = e.stats()
    assert got == {k: 2 * v for k, v in want.items()}
    assert st["counted"] == 2 * sum(want.values()) and st["insert_path"] >= len(want)


def test_arrow_export_frame():
    """finish() hands the frame over through the Arrow C Data Interface: schema {sequence: large_string,
    count: uint64}, same rows and order as the plain accessors, buffers stay valid after the context dies"""
    import pyarrow as pa
    reads = synth.make_reads(97, 30000, AD20, L=150, n_variants=3000)
    with _engine(AD20) as e:
        e.submit(reads)
        frame = e.finish()
        plain = list(oracle.sorted_table(e.finish_dict()))
        e.clear()
        empty = e.finish()
    tab = frame if isinstance(frame, pa.Table) else frame.to_arrow()
    assert tab.column_names == ["sequence", "count"]
    assert tab.schema.field("count").type == pa.uint64()
    assert tab.schema.field("sequence").type in (pa.large_string(), pa.string_view(), pa.string())
    got = list(zip((s.encode() for s in tab["sequence"].to_pylist()), tab["count"].to_pylist()))
    assert got == plain and len(got) > 500
    assert (empty.num_rows if isinstance(empty, pa.Table) else empty.height) == 0


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """bench.py's N>1 path (read sharding by rank, barrier/max timing, table export -> all-gather ->
    device merge, merged-count check) rehearsed with 2 ranks sharing cuda:0 over gloo; on the 8-GPU
    node the same code runs over RCCL"""
    import json
    import socket
    import subprocess
    import sys
    from tests.conftest import ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
            "--reads", "400000", "--variants", "30000", "--backend", "gloo", "--same-device", "--no-cpu-baseline"]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    # the way the driver starts N > 1 (torch.distributed.run around bench.py), and bench.py --gpus 2 on its own, which
    # must start the two ranks itself
    for cmd in (launcher + tail, [sys.executable] + tail):
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        j = json.loads(line)
        assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
        assert j["merge_ms"] is not None and j["parity_gate"] is True and j["merge_error"] is None


def test_merge_grows_the_receiving_table():
    """rank 0 of N merges its peers' tables into a table that was sized for its own keys: four exported tables of
    disjoint keys go into a 1024-slot context (vf_table_merge reserves room from the blob header, as a submit does);
    a blob whose header promises more than it holds is refused"""
    import torch
    from vfind_amd import Engine
    rng = np.random.default_rng(123)
    parts = []
    for k in range(4):                                   # disjoint variant libraries: different seeds, 3000 variants each
        parts.append(synth.make_reads(400 + k, 9000, AD20, L=150, n_variants=3000))
    with Engine(AD20, table_slots=1 << 10, arena_bytes=1 << 22) as dst:
        dst.submit(parts[0][:500])                       # a few keys of its own: the table stays at 1024 slots
        bufs = []
        for k in range(4):
            with Engine(AD20, table_slots=1 << 14, arena_bytes=1 << 22) as src:
                src.submit(parts[k])
                n = src.export_size()
                buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
                assert src.export_to(buf.data_ptr(), n + 64) == n
                bufs.append((buf, n))
        for buf, n in bufs:
            dst.merge_from(buf.data_ptr(), n)
        got = dst.finish_dict()
        assert dst.stats()["table_entries"] > 1 << 10
        # a header that claims more entries than the blob holds (u64 arithmetic must not wrap)
        bad = bufs[0][0].clone()
        bad[:8] = torch.tensor(list((1 << 61).to_bytes(8, "little")), dtype=torch.uint8)
        with pytest.raises(ValueError, match="truncated"):
            dst.merge_from(bad.data_ptr(), bufs[0][1])
    want = oracle.count(parts[0][:500] + [r for p in parts for r in p], AD20)
    assert got == want and len(got) > 4000
    # long keys (arena) through the same path
    long_a = synth.make_reads(410, 1500, AD20, L=600, n_variants=1500)
    long_b = synth.make_reads(411, 1500, AD20, L=600, n_variants=1500)
    with Engine(AD20, skip_translation=True, table_slots=1 << 8, arena_bytes=1 << 14) as dst, \
         Engine(AD20, skip_translation=True, table_slots=1 << 12, arena_bytes=1 << 22) as src:
        dst.submit(long_a[:50])
        src.submit(long_b)
        n = src.export_size()
        buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        src.export_to(buf.data_ptr(), n + 64)
        dst.merge_from(buf.data_ptr(), n)
        assert dst.finish_dict() == oracle.count(long_a[:50] + long_b, AD20, skip_translation=True)


def test_table_growth_keeps_long_keys_in_place():
    """keys longer than 56 bytes live in the arena; growing the table re-inserts every entry and must not copy those
    keys again (each growth used to double the arena's fill): 600-base regions, table started at 64 slots"""
    from vfind_amd import Engine
    reads = synth.make_reads(420, 6000, AD20, L=600, n_variants=6000)
    want = oracle.count(reads, AD20, skip_translation=True)
    with Engine(AD20, skip_translation=True, table_slots=64, arena_bytes=1 << 16, batch_reads=500) as e:
        e.submit(reads)
        got = e.finish_dict()
        st = e.stats()
    assert got == want and max(len(k) for k in got) > 224
    assert st["table_entries"] >= 2000                  # grew from 64 slots several times
    # arena handed out = the keys' own dwords plus chunk slack (one partial 1024-dword chunk per workgroup and stream),
    # not a multiple of them
    assert st["arena_top"] <= st["arena_dwords"] + 4 * 2048 * 1024, st
    assert st["arena_top"] < 2 * st["arena_dwords"] + (1 << 20), st


def test_two_files_into_one_context(tmp_path):
    """BASELINE configs[2] shape: R1 and R2 as two gzip FASTQ files counted into ONE table (run_fastq twice on one
    context) equals the oracle on the concatenation"""
    import vfind_amd
    r1 = synth.make_reads(430, 40000, AD20, L=150, sub=0.01, indel=0.002, n_variants=600)
    r2 = synth.make_reads(431, 30000, AD20, L=150, sub=0.01, indel=0.002, n_variants=600, dirty=0.01)
    p1, p2 = str(tmp_path / "r1.fq.gz"), str(tmp_path / "r2.fq.gz")
    synth.write_fastq_gz(p1, r1, members=4)
    synth.write_fastq_gz(p2, r2, members=1)
    with vfind_amd.Engine(AD20, n_threads=4, table_slots=1 << 14) as e:
        e.run_fastq(p1)
        e.run_fastq(p2)
        got = e.finish_dict()
        assert e.stats()["reads"] == len(r1) + len(r2)
    assert got == oracle.count(r1 + r2, AD20)


def test_counting_path_spans_follow_the_oracle():
    """the counting path skips the suffix alignment of a read whose prefix failed (a read without a start contributes
    nothing, lib.rs:288); its per-read coordinates must equal the oracle's wherever the oracle finds a start, and hold
    no start wherever the oracle finds none"""
    import torch
    from vfind_amd import Engine
    from benchlib import synth as vsynth
    for seed, sub, indel in ((0x5EED0002, 0.005, 0.0005), (0x5EED0004, 0.15, 0.0), (77, 0.04, 0.02)):
        ads = vsynth.adapters(seed, 20)
        gen = vsynth.Generator(seed, ads, n_variants=20000, sub_rate=sub, indel_rate=indel)
        n = 200_000
        rows, lens = gen.packed(0, n, 8)
        data, offs = gen.ascii(0, n, 8)
        d_rows = torch.from_numpy(rows).cuda()
        d_lens = torch.from_numpy(lens.view("int16")).cuda()
        d_spans = torch.empty((n, 2), dtype=torch.int16, device="cuda")
        with Engine(ads, table_slots=1 << 17) as e:
            e.process_device(d_rows.data_ptr(), rows.shape[1], d_lens.data_ptr(), n, d_spans.data_ptr())
            got = e.finish_dict()
        sp = d_spans.cpu().numpy().astype(np.int64)
        ost, oen = oracle.spans((data, offs), ads, n_threads=8)
        has = ost >= 0
        assert (sp[has, 0] == ost[has]).all() and (sp[has, 1] == np.maximum(oen[has], -1)).all()
        assert (sp[~has, 0] < 0).all()
        assert has.sum() > n // 10 and (~has).sum() > 1000
        assert got == oracle.count((data, offs), ads, n_threads=8)


@pytest.mark.parametrize("name,kw,n", [
    ("cfg4", dict(seed=0x5EED0004, read_len=150, m=20, sub_rate=0.15, indel_rate=0.0, n_variants=1_000_000), 1_000_000),
    ("cfg5", dict(seed=0x5EED0005, read_len=300, m=30, sub_rate=0.005, indel_rate=0.0005, n_variants=10_000_000), 1_000_000)])
def test_baseline_config_shapes_million_reads(name, kw, n):
    """BASELINE.json configs[3] and [4] shapes (15 % adapter substitutions: an alignment on nearly every read; 300 bp reads,
    30 bp adapters, 10^7-variant library) at 10^6 reads through the resident path: the whole table equals the oracle's"""
    import torch
    from vfind_amd import Engine
    from benchlib import synth as vsynth
    ads = vsynth.adapters(kw["seed"], kw["m"])
    gen = vsynth.Generator(kw["seed"], ads, read_len=kw["read_len"], n_variants=kw["n_variants"], sub_rate=kw["sub_rate"], indel_rate=kw["indel_rate"])
    rows, lens = gen.packed(0, n, 8)
    data, offs = gen.ascii(0, n, 8)
    want = oracle.count((data, offs), ads, n_threads=16)
    d_rows = torch.from_numpy(rows).cuda()
    d_lens = torch.from_numpy(lens.view("int16")).cuda()
    with Engine(ads, table_slots=1 << 21) as e:
        e.process_device(d_rows.data_ptr(), rows.shape[1], d_lens.data_ptr(), n)
        got = e.finish_dict()
        st = e.stats()
    assert got == want and len(got) > 5000
    assert st["dp_prefix"] > (500_000 if name == "cfg4" else 50_000)


def test_owner_partitioned_exchange_between_local_contexts():
    """the multi-GPU reduce as the library offers it without a transport of its own: every context buckets its entries by
    owner (vf_table_partition_sizes / _export), bucket o goes to context o, every owner clears and merges what it holds,
    the owners' shares are gathered at the root -- three contexts on one GPU, loop-back transport (device memory)"""
    import torch
    from vfind_amd import Engine
    from vfind_amd.dist import allreduce_local
    parts = [synth.make_reads(500 + k, 12000, AD20, L=150, n_variants=4000, sub=0.01) for k in range(3)]
    engines = [Engine(AD20, table_slots=1 << (10 + 2 * k), arena_bytes=1 << 22) for k in range(3)]
    try:
        for e, p in zip(engines, parts):
            e.submit(p)
        # the shares of one table are disjoint and complete: their frames add up to the table's own frame
        sizes = engines[1].partition_sizes(3)
        assert len(sizes) == 3 and all(s >= 16 for s in sizes)
        shares = []
        for o in range(3):
            buf = torch.empty(sizes[o] + 64, dtype=torch.uint8, device="cuda")
            assert engines[1].partition_export(3, o, buf.data_ptr(), buf.numel()) == sizes[o]
            shares.append((buf, sizes[o]))
        own = engines[1].finish_dict()
        per_share = [engines[1].finish_blobs_dict([(b.data_ptr(), n)]) for b, n in shares]
        assert sum(len(d) for d in per_share) == len(own) and min(len(d) for d in per_share) > len(own) // 6
        merged = {}
        for d in per_share:
            assert not (merged.keys() & d.keys())
            merged.update(d)
        assert merged == own == oracle.count(parts[1], AD20)
        allreduce_local(engines, root=0)
        got = engines[0].finish_dict()
        rest = [e.finish_dict() for e in engines[1:]]
    finally:
        for e in engines:
            e.close()
    want = oracle.count([r for p in parts for r in p], AD20)
    assert got == want and len(got) > 5000
    # the other contexts are left with their own share of the sum
    for d in rest:
        assert 0 < len(d) < len(want) and all(want[k] == v for k, v in d.items())


def test_frame_from_several_exported_tables_sums_equal_strings():
    """vf_finish_blobs: the tables of several contexts (overlapping key sets, translated and untranslated, long keys) laid end
    to end and summed by the device result build -- same rows, counts and order as one context that saw all the reads"""
    import torch
    from vfind_amd import Engine
    a = synth.make_reads(510, 9000, AD20, L=150, n_variants=1500, dirty=0.02)
    b = synth.make_reads(510, 7000, AD20, L=150, n_variants=1500, dirty=0.02)[::-1]      # the same library: mostly shared keys
    c = synth.make_reads(511, 2000, AD20, L=600, n_variants=300)                          # keys longer than 56 bytes
    for kw in (dict(), dict(skip_translation=True)):
        blobs, keep = [], []
        with Engine(AD20, table_slots=1 << 13, **kw) as e1, Engine(AD20, table_slots=1 << 15, **kw) as e2, Engine(AD20, table_slots=1 << 12, **kw) as e3:
            for e, reads in ((e1, a), (e2, b), (e3, c)):
                e.submit(reads)
                n = e.export_size()
                buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
                e.export_to(buf.data_ptr(), buf.numel())
                keep.append(buf)
                blobs.append((buf.data_ptr(), n))
            got = e1.finish_blobs_dict(blobs + [blobs[0]])                                  # the first table twice
            assert e1.finish_dict() == oracle.count(a, AD20, **kw)                         # e1's own table is untouched
        want = oracle.count(a + a + b + c, AD20, **kw)
        assert got == want and len(got) > 1000


@pytest.mark.parametrize("world,root", [(2, 0), (3, 2), (8, 5)])
def test_table_allreduce_between_local_contexts(world, root):
    """vf_table_allreduce itself - bucket export, send / receive / gather offsets, the round of transfers, the share gather,
    the merges - with world > 1: `world` contexts on this one device, one thread each, joined by the in-process transport
    (RCCL refuses two ranks on one device; only the five transport calls differ from the RCCL path).  Root ends up with the
    oracle's table of ALL reads, every other rank with its own share: disjoint key sets whose union is that table"""
    from vfind_amd import Engine, LocalGroup
    parts = [synth.make_reads(700 + r, 2500 + 300 * r, AD20, L=150, n_variants=600, sub=0.004, indel=0.0) for r in range(world)]
    want = oracle.count([x for p in parts for x in p], AD20)
    engines = [Engine(AD20, table_slots=1 << 12, arena_bytes=1 << 22) for _ in range(world)]      # (small tables: the merges must grow them)
    try:
        for e, p in zip(engines, parts):
            e.submit(p)
        g = LocalGroup(world)
        errs = g.reduce(engines, root=root)
        assert errs == [None] * world, errs
        tables = [e.finish_dict() for e in engines]
        assert tables[root] == want
        seen = {}
        for r, t in enumerate(tables):
            if r == root:
                continue
            for k, v in t.items():
                assert k not in seen, "shares overlap"
                assert want[k] == v
                seen[k] = v
        # root's own share is the rest
        assert set(seen) <= set(want)
        # a second exchange on the same group (the rendezvous is reusable): root's table doubles for the keys it owned...
        # just check that it completes and root still holds every key
        errs = g.reduce(engines, root=root)
        assert errs == [None] * world, errs
        assert set(engines[root].finish_dict()) == set(want)
        g.close()
    finally:
        for e in engines:
            e.close()


def test_table_allreduce_every_rank_leaves_when_one_fails(monkeypatch):
    """a rank that fails locally between the collectives (round 2 returned early there and left its peers in a receive) says so
    in the next status word: every rank comes back with an error, at every step where a rank can fail"""
    import vfind_amd
    from vfind_amd import Engine, LocalGroup
    world = 3
    parts = [synth.make_reads(800 + r, 2000, AD20, L=150, n_variants=300) for r in range(world)]
    for step in (1, 2, 3):
        engines = [Engine(AD20, table_slots=1 << 14, arena_bytes=1 << 22) for _ in range(world)]
        try:
            for e, p in zip(engines, parts):
                e.submit(p)
            monkeypatch.setenv("VFIND_TEST_FAIL_ALLREDUCE", f"1:{step}")
            g = LocalGroup(world)
            errs = g.reduce(engines, root=0)
            monkeypatch.delenv("VFIND_TEST_FAIL_ALLREDUCE")
            assert all(e is not None for e in errs), (step, errs)
            assert isinstance(errs[1], MemoryError) and "injected" in str(errs[1])
            assert "rank 1 failed" in str(errs[0]) and "rank 1 failed" in str(errs[2])
            g.close()
        finally:
            for e in engines:
                e.close()
    # ... and INSIDE a round of transfers, between begin() and end() (on RCCL: between ncclGroupStart and ncclGroupEnd, where an
    # early return used to leave the group open and the peers in their ncclGroupEnd): the failing rank closes the round and
    # breaks the communicator, its peers leave the rendezvous with an error instead of waiting for a send that never comes
    engines = [Engine(AD20, table_slots=1 << 14, arena_bytes=1 << 22) for _ in range(world)]
    try:
        for e, p in zip(engines, parts):
            e.submit(p)
        monkeypatch.setenv("VFIND_TEST_FAIL_ALLREDUCE", "1:4")
        g = LocalGroup(world)
        errs = g.reduce(engines, root=0)
        monkeypatch.delenv("VFIND_TEST_FAIL_ALLREDUCE")
        assert all(e is not None for e in errs), errs
        assert "inside a round" in str(errs[1])
        assert "a peer left the exchange" in str(errs[0]) and "a peer left the exchange" in str(errs[2])
        g.close()
        # the contexts are still usable, and a new group works
        g2 = LocalGroup(world)
        assert g2.reduce(engines, root=0) == [None] * world
        assert engines[0].finish_dict() == oracle.count([x for p in parts for x in p], AD20)
        g2.close()
    finally:
        for e in engines:
            e.close()


def test_rccl_entry_points_single_rank():
    """vf_comm_unique_id / vf_comm_init / vf_table_allreduce through the C ABI with a communicator of one rank (RCCL refuses
    two ranks on one GPU, so more cannot run on this box): the library loads librccl, creates and destroys a communicator,
    and the reduce of a one-rank world leaves the table as it is"""
    from vfind_amd import Engine
    reads = synth.make_reads(520, 5000, AD20, L=150, n_variants=500)
    with Engine(AD20, table_slots=1 << 12) as e:
        e.submit(reads)
        uid = Engine.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        e.comm_init(0, 1, uid)
        e.allreduce_rccl(0)
        assert e.finish_dict() == oracle.count(reads, AD20)
        with pytest.raises(ValueError):
            e.comm_init(2, 2, uid)


def test_rccl_transport_walks_the_whole_exchange_with_one_rank(monkeypatch):
    """RcclTransport itself - ncclAllGather of the status words, ncclGroupStart / ncclGroupEnd rounds, the polled waits with
    ncclCommGetAsyncError - through vf_table_allreduce with a one-rank communicator (test hook VFIND_TEST_ALLREDUCE_SINGLE; RCCL
    refuses two ranks on one GPU): the table goes through bucket export, clear, merge and share export and comes out as it was.
    And a deadline that cannot be met ends in an error and an aborted communicator, not in a hang."""
    from vfind_amd import Engine
    reads = synth.make_reads(521, 6000, AD20, L=150, n_variants=700)
    want = oracle.count(reads, AD20)
    monkeypatch.setenv("VFIND_TEST_ALLREDUCE_SINGLE", "1")
    with Engine(AD20, table_slots=1 << 12) as e:
        e.submit(reads)
        e.comm_init(0, 1, Engine.comm_unique_id())
        e.allreduce_rccl(0)
        assert e.finish_dict() == want
        e.allreduce_rccl(0)                                    # the communicator is reusable
        assert e.finish_dict() == want
        assert e.stats()["dup_rows"] == 0


def test_pinned_batch_api_round_trips_the_slots():
    """vf_batch_acquire / vf_batch_submit: the caller packs straight into the library's page-locked staging; one thread walks
    round the queue_len slots (acquire hands out the slot used longest ago), results equal the oracle, an acquired batch can be
    given back unused, over-long reads and over-full submits are refused"""
    from vfind_amd import Engine
    from benchlib import synth as vsynth
    ads = vsynth.adapters(77, 20)
    gen = vsynth.Generator(77, ads, n_variants=3000)
    n, nb = 40000, 5000
    data, offs = gen.ascii(0, n, 4)
    with Engine(ads, queue_len=3, table_slots=1 << 14) as e:
        seen = set()
        for k in range(n // nb):
            h, rows, lens = e.batch_acquire(nb + 100, gen.row_dwords)        # room for more than is submitted
            seen.add(rows.ctypes.data)
            gen.packed_into(k * nb, nb, rows.ctypes.data, lens.ctypes.data, 2)
            e.batch_submit(h, nb)
        assert len(seen) == 3                                                # all three slots took their turn
        h, rows, lens = e.batch_acquire(10, gen.row_dwords)
        e.batch_release(h)
        h, rows, lens = e.batch_acquire(10, gen.row_dwords)
        lens[:] = 150
        lens[3] = 161                                                        # does not fit a 10-word row
        with pytest.raises(ValueError, match="exceeds row"):
            e.batch_submit(h, 10)
        h, rows, lens = e.batch_acquire(10, gen.row_dwords)
        with pytest.raises(ValueError, match="more reads"):
            e.batch_submit(h, 11)
        # a handle is honoured once: the refused submit above gave the slot back, so the same handle is stale now - submitting
        # or releasing it again, or handing it to another context, is refused without touching any slot
        with pytest.raises(ValueError, match="not a batch handle"):
            e.batch_submit(h, 1)
        with pytest.raises(ValueError, match="not a batch handle"):
            e.batch_release(h)
        h2, rows, lens = e.batch_acquire(10, gen.row_dwords)
        with Engine(ads, table_slots=1 << 10) as other:
            with pytest.raises(ValueError, match="not a batch handle"):
                other.batch_submit(h2, 1)
        e.batch_release(h2)
        assert e.finish_dict() == oracle.count((data, offs), ads, n_threads=8)
        assert e.stats()["reads"] == n


def test_fuzz_parity_sample():
    """tests/tools/fuzz_parity.py in the suite: random adapter pairs (1-32 bp), scoring schemes (also ones that switch the
    filters off or leave the 2-bit path), thresholds, read lengths, error rates, dirty reads, both key kinds - spans and
    tables against the oracle, and no duplicate table entry.  A sample of the tool's space, seeded; DESIGN.md section 2
    records the long runs"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad = fuzz.run(90, 20261005, n_reads=1200, verbose=False)
    assert not bad, bad[:3]


def test_blob_header_is_read_in_stream_order():
    """vf_table_merge / vf_finish_blobs read the blob's 16-byte header behind whatever vf_wait_stream made a dependency of it:
    here the blob is written by a copy queued on a NON-BLOCKING stream of the caller behind a long-running fill, so at the time
    of the call the buffer still holds zeros (entry count 0: a blocking null-stream read of the header would merge nothing,
    silently).  ADVICE round 3, medium."""
    import torch
    from vfind_amd import Engine
    reads = synth.make_reads(530, 20000, AD20, L=150, n_variants=2000)
    want = oracle.count(reads, AD20)
    with Engine(AD20, table_slots=1 << 14) as src, Engine(AD20, table_slots=1 << 14) as dst, Engine(AD20, table_slots=1 << 14) as fr:
        src.submit(reads)
        n = src.export_size()
        blob = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        src.export_to(blob.data_ptr(), blob.numel())
        torch.cuda.synchronize()
        side = torch.cuda.Stream()                                   # hipStreamNonBlocking: not ordered with the null stream
        big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        for target, use in ((dst, "merge"), (fr, "blobs")):
            late = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                for _ in range(6):
                    big.fill_(1)                                     # ~ a millisecond of work in front of the copy
                late.copy_(blob, non_blocking=True)
            target.wait_stream(side.cuda_stream)
            if use == "merge":
                target.merge_from(late.data_ptr(), n)
                assert target.finish_dict() == want
            else:
                assert target.finish_blobs_dict([(late.data_ptr(), n)]) == want
            torch.cuda.synchronize()


def test_merge_refuses_byte_keys_a_tagged_table_cannot_hold():
    """a skip_translation context keeps regions with non-ASCII bytes as byte keys; its export merged into a TRANSLATING context
    (letter keys, cleared by epoch: the top bit of every key byte belongs to the epoch tag) must be refused loudly, not stored
    with those bits stripped.  ADVICE round 3."""
    from vfind_amd import Engine
    import torch
    ad = ("ACGTACGTAC", "TGCATGCATG")
    reads = [b"ACGTACGTAC" + "é".encode() * 3 + b"GATTACA" + b"TGCATGCATG", b"ACGTACGTAC" + b"GATTACAGA" + b"TGCATGCATG"]
    with Engine(ad, skip_translation=True, table_slots=1 << 10) as src, Engine(ad, table_slots=1 << 10) as dst:
        src.submit(reads * 3)
        assert src.finish_dict() == oracle.count(reads * 3, ad, skip_translation=True)
        n = src.export_size()
        blob = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        src.export_to(blob.data_ptr(), blob.numel())
        with pytest.raises(ValueError, match="non-ASCII"):
            dst.merge_from(blob.data_ptr(), n)


def test_unit_distance_scan_keeps_its_coverage():
    """VFIND_INDEL_SCAN=0: the unit-distance (Myers) scan for every scoring scheme - what every scheme got before round 4 and what
    schemes with cheap substitutions still get.  The default scheme now takes the indel scan, so the suite would otherwise reach
    the Myers path only through the fuzz's random schemes: here the shapes that stress it (15 % substitutions, indels, every read
    length mod 16) run through it with the defaults - spans and table against the oracle.  (Read once per process: a child.)"""
    import subprocess
    import sys
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import vfind_amd
from oracle import oracle
from tests import synth
AD20 = ("GATCGGAAGAGCACACGTCT", "AGATCGGAAGAGCGTCGTGT")
tot = 0
for seed, L, sub, indel in ((501, 150, 0.15, 0.0), (502, 151, 0.05, 0.03), (503, 157, 0.15, 0.0), (504, 160, 0.02, 0.01), (505, 300, 0.1, 0.01)):
    rd = synth.make_reads(seed, 6000, AD20, L=L, sub=sub, indel=indel, flank_max=9, vary_len=True)
    with vfind_amd.Engine(AD20, table_slots=1 << 15) as e:
        st, en = e.spans(rd)
        e.submit(rd)
        got = e.finish_dict()
        dup = e.stats()["dup_rows"]
    ost, oen = oracle.spans(rd, AD20)
    assert (st == ost).all() and (en == oen).all(), (seed, L)
    assert got == oracle.count(rd, AD20) and dup == 0, (seed, L)
    tot += len(got)
print("unit scan ok", tot)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, VFIND_INDEL_SCAN="0"))
    assert out.returncode == 0 and "unit scan ok" in out.stdout, out.stderr[-3000:]


def test_resident_batch_cut_into_pieces():
    """VFIND_RES_SPLIT: a resident batch (vf_process_device_packed) cut into pieces that go down the batch lanes one after the
    other (also what a batch of more than 2^30 reads gets: task entries are 31-bit) - spans of every piece land at their reads'
    places, the table equals the oracle's, no duplicate entries.  (The switch is read once per process: a child process.)"""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
from vfind_amd import Engine
from benchlib import synth as vsynth
from oracle import oracle
ads = vsynth.adapters(91, 20)
gen = vsynth.Generator(91, ads, n_variants=30000, sub_rate=0.03, indel_rate=0.004)
n = 300_000
rows, lens = gen.packed(0, n, 8)
data, offs = gen.ascii(0, n, 8)
d_rows = torch.from_numpy(rows).cuda(); d_lens = torch.from_numpy(lens.view("int16")).cuda()
d_spans = torch.full((n, 2), -7, dtype=torch.int16, device="cuda")
with Engine(ads, table_slots=1 << 17) as e:
    e.process_device(d_rows.data_ptr(), rows.shape[1], d_lens.data_ptr(), n, d_spans.data_ptr())
    got = e.finish_dict()
    st = e.stats()
sp = d_spans.cpu().numpy().astype(np.int64)
ost, oen = oracle.spans((data, offs), ads, n_threads=8)
has = ost >= 0
assert (sp[has, 0] == ost[has]).all() and (sp[has, 1] == np.maximum(oen[has], -1)).all()
assert (sp != -7).all()                      # every piece wrote its reads' spans
assert got == oracle.count((data, offs), ads, n_threads=8)
assert st["dup_rows"] == 0 and st["launches_scan"] == 0 and st["reads"] == n
print("pieces ok", len(got))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, VFIND_RES_SPLIT="70000"))
    assert out.returncode == 0 and "pieces ok" in out.stdout, out.stderr[-3000:]
