"""Pure-Python restatement of the semi-global DP -- TEST INFRASTRUCTURE ONLY (small cases).

Two things the C oracle cannot give:
  * sg_stats_py     -- a second, full-matrix implementation of SURVEY App. B.2 + tie policy P,
                       written independently of oracle/vf_oracle.c (which is column-rolling),
                       to cross-check it (the reference call it restates: src/lib.rs:155-160).
  * cooptimal       -- the set of ALL (end cell, length) pairs reachable by ANY optimal traceback,
                       i.e. what the `length` statistic could be under any tie policy.  A read is
                       "tie-sensitive" when that set holds more than one length (App. B.4); parity
                       with the real parasail is pinned only on tie-free reads.
"""
from __future__ import annotations

NEG = -(1 << 28)


def _code(c: int) -> int:
    return {65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3}.get(c, 4)


def _sub(a: int, b: int, match: int, mismatch: int) -> int:
    if a == 4 or b == 4:
        return 0
    return match if a == b else mismatch


def _matrices(q: bytes, r: bytes, match, mismatch, gopen, gext):
    m, n = len(q), len(r)
    H = [[0] * (n + 1) for _ in range(m + 1)]
    E = [[NEG] * (n + 1) for _ in range(m + 1)]
    F = [[NEG] * (n + 1) for _ in range(m + 1)]
    D = [[NEG] * (n + 1) for _ in range(m + 1)]
    qc = [_code(c) for c in q]
    rc = [_code(c) for c in r]
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            E[i][j] = max(H[i][j - 1] - gopen, E[i][j - 1] - gext)
            F[i][j] = max(H[i - 1][j] - gopen, F[i - 1][j] - gext)
            D[i][j] = H[i - 1][j - 1] + _sub(qc[i - 1], rc[j - 1], match, mismatch)
            H[i][j] = max(D[i][j], E[i][j], F[i][j])
    return H, E, F, D


def sg_stats_py(q, r, match=3, mismatch=-2, gopen=5, gext=2):
    """(score, length, end_query, end_ref) under tie policy P; 1-based end coordinates"""
    q, r = bytes(q), bytes(r)
    m, n = len(q), len(r)
    if m == 0 or n == 0:
        return (NEG, 0, 0, 0)
    H, E, F, D = _matrices(q, r, match, mismatch, gopen, gext)
    HL = [[0] * (n + 1) for _ in range(m + 1)]
    EL = [[0] * (n + 1) for _ in range(m + 1)]
    FL = [[0] * (n + 1) for _ in range(m + 1)]
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            # P1: open only if strictly greater than extend
            EL[i][j] = (HL[i][j - 1] if H[i][j - 1] - gopen > E[i][j - 1] - gext else EL[i][j - 1]) + 1
            FL[i][j] = (HL[i - 1][j] if H[i - 1][j] - gopen > F[i - 1][j] - gext else FL[i - 1][j]) + 1
            # P2: diagonal, then F (north), then E (west)
            if H[i][j] == D[i][j]:
                HL[i][j] = HL[i - 1][j - 1] + 1
            elif H[i][j] == F[i][j]:
                HL[i][j] = FL[i][j]
            else:
                HL[i][j] = EL[i][j]
    # P3: last row first (smallest column on ties), last column only if strictly greater
    best, bl, bq, bd = NEG, 0, 0, 0
    for j in range(1, n + 1):
        if H[m][j] > best:
            best, bl, bq, bd = H[m][j], HL[m][j], m, j
    for i in range(1, m):
        if H[i][n] > best:
            best, bl, bq, bd = H[i][n], HL[i][n], i, n
    return (best, bl, bq, bd)


def cooptimal(q, r, match=3, mismatch=-2, gopen=5, gext=2):
    """(score, {(end_query, end_ref, length), ...}) over every optimal end cell and traceback"""
    q, r = bytes(q), bytes(r)
    m, n = len(q), len(r)
    if m == 0 or n == 0:
        return NEG, set()
    H, E, F, D = _matrices(q, r, match, mismatch, gopen, gext)
    zero = frozenset([0])
    HS = [[zero] * (n + 1) for _ in range(m + 1)]
    ES = [[frozenset()] * (n + 1) for _ in range(m + 1)]
    FS = [[frozenset()] * (n + 1) for _ in range(m + 1)]
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            s = set()
            if H[i][j - 1] - gopen == E[i][j]:
                s |= {x + 1 for x in HS[i][j - 1]}
            if E[i][j - 1] - gext == E[i][j]:
                s |= {x + 1 for x in ES[i][j - 1]}
            ES[i][j] = frozenset(s)
            s = set()
            if H[i - 1][j] - gopen == F[i][j]:
                s |= {x + 1 for x in HS[i - 1][j]}
            if F[i - 1][j] - gext == F[i][j]:
                s |= {x + 1 for x in FS[i - 1][j]}
            FS[i][j] = frozenset(s)
            s = set()
            if D[i][j] == H[i][j]:
                s |= {x + 1 for x in HS[i - 1][j - 1]}
            if E[i][j] == H[i][j]:
                s |= ES[i][j]
            if F[i][j] == H[i][j]:
                s |= FS[i][j]
            HS[i][j] = frozenset(s)
    best = max(max(H[m][1:]), max([H[i][n] for i in range(1, m)] or [NEG]))
    ends = set()
    for j in range(1, n + 1):
        if H[m][j] == best:
            ends |= {(m, j, length) for length in HS[m][j]}
    for i in range(1, m):
        if H[i][n] == best:
            ends |= {(i, n, length) for length in HS[i][n]}
    return best, ends


def tie_sensitive(q, r, **kw) -> bool:
    _, ends = cooptimal(q, r, **kw)
    return len({e[2] for e in ends}) > 1
