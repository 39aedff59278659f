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
