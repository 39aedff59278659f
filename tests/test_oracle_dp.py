"""The C oracle's column-rolling DP against the independent full-matrix pure-Python restatement
(oracle/pyref.py), including indel-heavy pairs, wildcard bytes, lowercase, extreme lengths and other
scoring schemes; and the co-optimal enumerator's contract (policy P's answer is always one of the
co-optimal answers, the score is policy independent)."""
import numpy as np
import pytest

from oracle import oracle, pyref
from tests import synth


def _pairs(seed, n, m_lo=1, m_hi=32, n_lo=0, n_hi=90, dirty=False):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        m = int(rng.integers(m_lo, m_hi + 1))
        q = synth.rand_seq(rng, m)
        r = synth.rand_seq(rng, int(rng.integers(0, 12))) + synth.mutate(rng, q, 0.08, 0.08) + synth.rand_seq(rng, int(rng.integers(0, 40)))
        r = r[: int(rng.integers(n_lo, n_hi + 1))]
        if dirty and rng.random() < 0.5 and r:
            b = bytearray(r)
            b[int(rng.integers(0, len(b)))] = ord("N") if rng.random() < 0.5 else ord("t")
            r = bytes(b)
        out.append((q, r))
    return out


@pytest.mark.parametrize("scheme", [(3, -2, 5, 2), (1, -1, 2, 1), (5, -4, 10, 1), (2, -3, 0, 0), (4, -1, 3, 3)])
def test_c_oracle_equals_python_restatement(scheme):
    ma, mi, go, ge = scheme
    for q, r in _pairs(hash(scheme) & 0xFFFF, 250, dirty=True):
        want = pyref.sg_stats_py(q, r, ma, mi, go, ge)
        got = oracle.sg_stats(q, r, ma, mi, go, ge)
        if len(r) == 0:
            assert got[0] < -10**8 and want[0] < -10**8
        else:
            assert got == want, (q, r)


def test_policy_answer_is_cooptimal_and_score_is_policy_free():
    n_tie = 0
    for q, r in _pairs(77, 300, m_lo=6, m_hi=24, n_lo=10):
        score, length, eq, ed = oracle.sg_stats(q, r)
        best, ends = pyref.cooptimal(q, r)
        assert score == best
        assert (eq, ed, length) in ends
        n_tie += len({e[2] for e in ends}) > 1
    assert n_tie > 0


def test_wildcards_and_case():
    # every byte outside ACGT (either case) scores 0 against everything; lowercase aligns like uppercase
    assert oracle.sg_stats("ACGT", "ACGT")[0] == 12
    assert oracle.sg_stats("ACGT", "acgt")[0] == 12
    assert oracle.sg_stats("ACNT", "ACGT")[0] == 9
    assert oracle.sg_stats("ACGT", "ACNT")[0] == 9
    assert oracle.sg_stats("NNNN", "ACGT")[0] == 0
    assert oracle.sg_stats("ACGU", "ACGT")[0] == 9          # U is not in the aligner's alphabet (lib.rs:236)


def test_find_adapter_match_contract():
    # exact hit wins over any alignment; prefix returns k+m, suffix k (lib.rs:150-153)
    st, en = oracle.spans([b"TTACGTACGTAAGGGCCCTT"], ("ACGTACGT", "GGGCCC"))
    assert (int(st[0]), int(en[0])) == (10, 12)
    # alignment fallback: prefix cut = alignment length, suffix cut = n - length (lib.rs:159-160)
    read = b"ACGTACGAACGTTTTTTTTTTTTGGGCACCCAA"
    st, en = oracle.spans([read], ("ACGTACGTACGT", "GGGCCCCCAA"), accept_prefix_alignment=0.6, accept_suffix_alignment=0.6)
    sp = oracle.sg_stats("ACGTACGTACGT", read)
    ss = oracle.sg_stats("GGGCCCCCAA", read)
    assert int(st[0]) == sp[1] and int(en[0]) == len(read) - ss[1]
