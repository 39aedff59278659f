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


def test_indel_scan_bit_parallel_equals_the_recurrence():
    """round 4 (vf_k_front.hip indel_column / myers_scan<W, true>): the two-lattice bit-parallel scan gives the indel-distance
    search row C[m][j] of the plain recurrence - adapters of every length 1..32, reads with planted noisy copies, two- and
    four-letter alphabets (long runs of equal deltas are where a carry chain would break)"""
    rng = random.Random(20261007)
    for it in range(1500):
        m = rng.randint(1, 32)
        k = 2 if it % 3 == 0 else 4
        ad = [rng.randrange(k) for _ in range(m)]
        cp = list(ad)
        for _ in range(rng.randint(0, 5)):
            p = rng.randrange(len(cp)) if cp else 0
            t = rng.random()
            if t < 0.4 and cp:
                cp[p] = rng.randrange(k)
            elif t < 0.7 and cp:
                del cp[p]
            else:
                cp.insert(p, rng.randrange(k))
        if it % 4 == 0:
            cp = []
        read = [rng.randrange(k) for _ in range(rng.randint(0, 20))] + cp + [rng.randrange(k) for _ in range(rng.randint(0, 40))]
        read = read or [0]
        assert band_model.indel_last_row_bitparallel(ad, read) == band_model.indel_last_row(ad, read)


def test_indel_scan_never_turns_an_acceptable_task_away():
    """the filter property ("the indel scan"): under every scheme for which the host picks the indel scan, an
    alignment the full-matrix DP accepts ends in a column with C <= kthr = max_indel_units(T).  Random schemes, 0-5 edits
    (substitutions, insertion runs, deletions), cut reads; the check bites with a threshold one unit short; and the scan does
    turn away tasks the unit-distance scan would have sent to the DP (what it is for)"""
    rng = random.Random(20261008)
    stats, seen = {}, 0
    while seen < 1200:
        good = band_model.scan_case(rng, stats)
        if good is None:
            continue
        assert good
        seen += 1
    assert stats.get("accepted", 0) > 300 and stats.get("rejected", 0) > 300
    assert stats.get("turned away (the unit scan would have passed it on)", 0) > 20
    import contextlib
    import io
    wrong = seen = 0
    with contextlib.redirect_stdout(io.StringIO()):
        while seen < 600:
            good = band_model.scan_case(rng, None, 1)
            if good is None:
                continue
            seen += 1
            wrong += not good
    assert wrong > 0


def test_host_bounds_of_the_default_scheme():
    """vfind's defaults (3 / -2 / 5 / 2, accept 0.75, 20 bp adapter: T = 46, loss <= 14): five unit edits (a five-base insertion)
    and five indel units (the same, or a substitution and a three-base insertion); three substitutions are 6 > 5"""
    assert band_model.max_unit_errors(46, 20, 3, -2, 5, 2) == 5
    assert band_model.max_indel_units(46, 20, 3, -2, 5, 2) == 5
    assert band_model.max_indel_units(55, 20, 3, -2, 5, 2) == 2 and band_model.max_indel_units(50, 20, 3, -2, 5, 2) == 4
