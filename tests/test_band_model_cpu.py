"""CPU check of the argument behind the filter's class 0 and the banded DP (dp_banded in k_dpall): the Python
restatement of the kernel's control flow (tests/tools/band_model.py) must agree with the full-matrix DP of
oracle/pyref.py (reference call: src/lib.rs:155-160) on random class-0 tasks -- and stop agreeing when the band is
put in the wrong place, so the check is known to bite."""
import random

from tests.tools import band_model


def test_banded_dp_equals_full_dp_on_class0_tasks():
    rng = random.Random(20261004)
    ran = 0
    while ran < 250:
        good = band_model.one_case(rng)
        if good is None:
            continue
        ran += 1
        assert good


def test_model_detects_a_misplaced_band(monkeypatch):
    real = band_model.banded
    monkeypatch.setattr(band_model, "banded", lambda q, r, av, *a: real(q, r, av + 3, *a))
    rng = random.Random(3)
    ran = bad = 0
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        while ran < 150:
            good = band_model.one_case(rng)
            if good is None:
                continue
            ran += 1
            bad += not good
    assert bad > 0


def test_block_filter_classes_equal_full_dp():
    """the block filter (k_bfilter): copies of kerr_lo + 1 disjoint adapter blocks, a diagonal certified
    at Hamming distance <= 1, band [dmin - K, dmax + K] of 5 or 10 diagonals -- restated in tests/tools/band_model.py
    and compared with the full-matrix DP on random tasks (substitutions, indels, flanks, cut reads, second copies,
    stray block copies, repetitive adapters, random scoring schemes)"""
    rng = random.Random(20261005)
    ran, stats = 0, {}
    while ran < 250:
        good = band_model.block_case(rng, stats)
        if good is None:
            continue
        ran += 1
        assert good
    assert stats.get(band_model.KBAND_N, 0) > 50 and stats.get(band_model.KBAND, 0) > 20


def test_block_model_detects_a_band_that_is_too_narrow(monkeypatch):
    """with K - 1 in the place of K (a band one diagonal short on either side) the model must disagree with the full DP
    on some tasks: gapped optimal alignments leave the narrower band"""
    real = band_model.block_filter

    def narrow(ad, read, K, plan, row_bases):
        got = real(ad, read, K, plan, row_bases)
        if got is None:
            return None
        kband, av = got
        return kband - 2, av + 1
    monkeypatch.setattr(band_model, "block_filter", narrow)
    rng = random.Random(11)
    ran = bad = 0
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        while ran < 400:
            good = band_model.block_case(rng)
            if good is None:
                continue
            ran += 1
            bad += not good
    assert bad > 0


def test_wide_band_class_equals_full_dp(monkeypatch):
    """class 6 of the Myers filter (vf_kernels_common.h kBandW, vf_k_front.hip myers_classify): a task whose detected end columns form ONE cluster [jmin, jmax]
    with jmax - jmin + 2 kerr + 1 <= 16 runs the banded DP on the diagonals [jmin - m - kerr, jmax - m + kerr], whatever its
    edits.  Restated on the CPU against the full-matrix DP: accept / reject, score and length agree; and the model bites -
    a band placed with kerr - 1 in the place of kerr loses alignments"""
    rng = random.Random(20261005)
    stats, seen = {}, 0
    while seen < 2500:
        good = band_model.wide_case(rng, stats)
        if good is None:
            continue
        assert good
        seen += 1
    assert stats.get("accepted", 0) > 500 and stats.get("rejected", 0) > 500
    real = band_model.max_unit_errors
    monkeypatch.setattr(band_model, "max_unit_errors", lambda *a: (lambda k: k if k is None or k < 2 else k - 1)(real(*a)))
    wrong = seen = 0
    while seen < 1500:
        good = band_model.wide_case(rng)
        if good is None:
            continue
        seen += 1
        wrong += not good
    assert wrong > 0


def test_hamming_certified_band_equals_full_dp(monkeypatch):
    """round 4 (vf_k_front.hip hamming_band_class): a single-cluster task whose cluster's centre diagonal carries the adapter
    with h substitutions has an optimum of at least match*m - h*(match - mismatch), so its optimal alignments have at most
    kq = max_unit_errors of that bound unit edits and lie on (hi - lo) + 2 kq + 1 diagonals: restated on the CPU against the
    full-matrix DP over random scoring schemes, 2-5 edits, flanks, cut reads, repetitive adapters and second copies - accept /
    reject, score and length agree.  The model bites: with the band two diagonals off it disagrees on ~4 % of the tasks (and with
    kq - 2 in the place of kq on ~1 %; kq - 1 survives random tasks - the bound has a unit of slack against them - which is why the
    proof, not the sample, is the argument).  A sample: the tool's 10 000-task run is profiles/r04_band_model_10000.log."""
    rng = random.Random(20261006)
    stats, seen = {}, 0
    while seen < 500:
        good = band_model.hamming_case(rng, stats)
        if good is None:
            continue
        assert good
        seen += 1
    assert stats.get("accepted", 0) > 400
    assert sum(v for k, v in stats.items() if isinstance(k, tuple) and k[0] == band_model.KBAND and k[1] == 2) > 100      # the two-substitution tasks
    assert sum(v for k, v in stats.items() if isinstance(k, tuple) and k[1] >= 3) > 10
    import contextlib
    import io
    real = band_model.banded
    monkeypatch.setattr(band_model, "banded", lambda q, r, av, *a: real(q, r, av + 2, *a))
    wrong = seen = 0
    with contextlib.redirect_stdout(io.StringIO()):
        while seen < 250:
            good = band_model.hamming_case(rng)
            if good is None:
                continue
            seen += 1
            wrong += not good
    assert wrong > 0
