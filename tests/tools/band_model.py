#!/usr/bin/env python3
"""CPU model of the filter's class-0 decision and of the banded DP (dp_banded, vf_k_dp.hip), checked against the
full-matrix DP of oracle/pyref.py on random tasks.  No GPU: it restates the kernel's control flow (relative
columns, rows per column, what a cell reads outside the band) in Python, so the argument of DESIGN.md 4.3
(tightened edit bound, shrunk cluster, band of diagonals) is tested on its own.

    python tests/tools/band_model.py [cases] [seed]
"""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import pyref  # noqa: E402  (test infrastructure)

NEG = -(1 << 28)
KBAND = 10


def max_unit_errors(T, m, match, mismatch, gopen, gext):
    L = match * m - T
    if L < 0:
        return -1
    if match < 1 or match - mismatch < 1 or gext < 1 or gopen < 0:
        return None
    c_other = min(match, match - mismatch, min(gopen, gext) + match)
    best = L // c_other
    g = 1
    while g * gopen <= L and g <= 4096:
        ln = g
        while g * gopen + (ln - g) * gext <= L:
            best = max(best, ln + (L - g * gopen - (ln - g) * gext) // c_other)
            ln += 1
        g += 1
    return best


def myers_last_row(q, r):
    m, n = len(q), len(r)
    prev = list(range(m + 1))
    out = [m]
    for j in range(1, n + 1):
        cur = [0] * (m + 1)
        for i in range(1, m + 1):
            cur[i] = min(prev[i - 1] + (q[i - 1] != r[j - 1]), prev[i] + 1, cur[i - 1] + 1)
        out.append(cur[m])
        prev = cur
    return out


def max_indel_units(T, m, match, mismatch, gopen, gext):
    """host: max_indel_units (vf_host.cpp) - the most INDEL units (a mismatch counts two) an alignment with score >= T can have;
    -1 nothing reaches T, None no (useful) bound"""
    L = match * m - T
    if L < 0:
        return -1
    if match < 1 or match - mismatch < 1 or gext < 1 or gopen < 1 or L > 2048:
        return None
    csub = match - mismatch
    c1 = min(match, min(gopen, gext) + match)
    fill = []
    for R in range(L + 1):
        b, sub = 0, 0
        while sub * csub <= R:
            b = max(b, 2 * sub + (R - sub * csub) // c1)
            sub += 1
        fill.append(b)
    best = fill[L]
    g = 1
    while g * gopen <= L:
        ln = g
        while g * gopen + (ln - g) * gext <= L:
            best = max(best, ln + fill[L - g * gopen - (ln - g) * gext])
            ln += 1
        g += 1
    return best


def indel_last_row(q, r):
    """C[m][j], j = 0..n, of the indel-distance search (free start in the read): the plain recurrence"""
    m, n = len(q), len(r)
    prev = list(range(m + 1))
    out = [m]
    for j in range(1, n + 1):
        cur = [0] * (m + 1)
        for i in range(1, m + 1):
            cur[i] = prev[i - 1] if q[i - 1] == r[j - 1] else 1 + min(prev[i], cur[i - 1])
        out.append(cur[m])
        prev = cur
    return out


def indel_last_row_bitparallel(q, r):
    """the kernel's two-lattice scan (vf_k_front.hip, indel_column / myers_scan<W, true>), 32-bit words, adapter in the top m bits"""
    m = len(q)
    up = 32 - m
    M32 = 0xFFFFFFFF
    eq = [0, 0, 0, 0]
    for i, c in enumerate(q):
        eq[c] |= 1 << (up + i)
    na = [0, 0]                                  # ~A of the even / odd lattice
    out = [m]
    for j in range(1, len(r) + 1):
        E = eq[r[j - 1]]
        best = None
        for lat in (0, 1):
            c0 = (j + lat) & 1                   # C[0][j] of the lattice; it steps up into this column <=> c0 == 1
            P = ~(na[lat] | E) & M32
            sm = ((P | na[lat]) + na[lat] + c0) & M32
            na[lat] = sm & ~P & M32
            C = c0 + m - 2 * bin(na[lat]).count("1")
            best = C if best is None else min(best, C)
        out.append(best)
    return out


METRIC = "auto"        # "unit": always the Myers scan; "auto": the indel scan where its bound is no larger than kerr (host: fill_dp)


class Scan:
    """what the pre-DP scan of one task sees, in the metric the host picks for the scheme: D[j] per end column, the threshold
    kthr, and shrink(T_tight, k_unit) = how far inside a cluster the end columns of alignments scoring >= T_tight lie"""
    def __init__(self, ad, read, T, kerr, match, mismatch, gopen, gext):
        m = len(ad)
        self.args = (m, match, mismatch, gopen, gext)
        ki = max_indel_units(T, m, match, mismatch, gopen, gext) if METRIC == "auto" else None
        self.indel = ki is not None and 1 <= ki <= kerr
        if self.indel:
            self.D, self.kthr = indel_last_row(ad, read), ki
        else:
            self.D, self.kthr = myers_last_row(ad, read), kerr
        self.kerr = kerr

    def shrink(self, T_tight, k_unit):
        if not self.indel:
            return self.kerr - k_unit
        m, match, mismatch, gopen, gext = self.args
        k = max_indel_units(T_tight, m, match, mismatch, gopen, gext)
        return self.kthr - k if (k is not None and 0 <= k <= self.kthr) else 0


def clusters(det_cols, gap):
    ncl, c = 0, [0, 0, 0, 0]
    for j in det_cols:
        if ncl == 0:
            ncl = 1; c[0] = c[1] = j
        elif ncl == 1:
            if j - c[1] > gap:
                ncl = 2; c[2] = c[3] = j
            else:
                c[1] = j
        elif ncl == 2:
            if j - c[3] > gap:
                ncl = 3
            else:
                c[3] = j
    return ncl, c


def banded(q, r, av, match, mismatch, gopen, gext, KBAND=KBAND):
    """dp_banded<M, B> restated (B = KBAND diagonals): cells are (score, length); returns (score, length)"""
    m, n = len(q), len(r)
    KT = m + KBAND - 1
    H = [(0, 0)] * m           # column 0 of the read
    E = [(NEG, 0)] * m
    k0 = 1 - av if av < 0 else 1
    kend = min(KT, n - av)
    best = (NEG, 0)
    done = None
    for k in range(1, KT + 1):
        if not (k0 <= k <= kend):
            continue
        j = av + k
        r0, r1 = max(1, k - KBAND + 1), min(m, k)
        hup = (0, 0) if r0 == 1 else (NEG, 0)
        f1 = (NEG, 0)
        hdiag = (0, 0) if r0 == 1 else H[r0 - 2]
        for row in range(r0, r1 + 1):
            hc = (NEG, 0) if row == k else H[row - 1]
            ec = (NEG, 0) if row == k else E[row - 1]
            eo, ee = hc[0] - gopen, ec[0] - gext
            e = (eo, hc[1] + 1) if eo > ee else (ee, ec[1] + 1)
            fo, fe = hup[0] - gopen, f1[0] - gext
            f = (fo, hup[1] + 1) if fo > fe else (fe, f1[1] + 1)
            a, b = q[row - 1], r[j - 1]
            d = (hdiag[0] + (match if a == b else mismatch), hdiag[1] + 1)
            hs = max(d[0], e[0], f[0])
            h = d if hs == d[0] else (f if hs == f[0] else e)
            hdiag = hc
            H[row - 1] = h
            E[row - 1] = e
            hup, f1 = h, f
        if r1 == m and H[m - 1][0] > best[0]:
            best = H[m - 1]
        if j == n:
            bs, bl = best
            for row in range(r0, min(r1, m - 1) + 1):
                if H[row - 1][0] > bs:
                    bs, bl = H[row - 1]
            done = (bs, bl)
    return done if done is not None else best


def one_case(rng):
    match = rng.choice([1, 2, 3, 3, 3, 4, 5])
    mismatch = rng.choice([-4, -3, -2, -2, -2, -1, 0])
    gopen = rng.choice([0, 2, 3, 5, 5, 5, 8])
    gext = rng.choice([1, 2, 2, 2, 3, 4])
    m = rng.randint(4, 24)
    acc = rng.choice([0.5, 0.6, 0.7, 0.75, 0.75, 0.8, 0.9])
    T = int(acc * match * m // 1) + 1
    ad = [rng.randrange(4) for _ in range(m)]
    # read: flank + adapter with a few edits + tail, sometimes cut at either end
    cp = list(ad)
    for _ in range(rng.choice([1, 1, 1, 1, 2, 2, 3])):
        t = rng.random()
        p = rng.randrange(len(cp))
        if t < 0.6:
            cp[p] = (cp[p] + rng.randint(1, 3)) % 4
        elif t < 0.8:
            cp.insert(p, rng.randrange(4))
        elif len(cp) > 2:
            del cp[p]
    flank = [rng.randrange(4) for _ in range(rng.choice([0, 0, 1, 2, 5, 9, 17]))]
    tail = [rng.randrange(4) for _ in range(rng.choice([0, 0, 1, 2, 3, 7, 20]))]
    read = flank + cp + tail
    if rng.random() < 0.2:
        read = read[rng.randint(1, 3):]
    if rng.random() < 0.2:
        read = read[: len(read) - rng.randint(1, 3)]
    if rng.random() < 0.1:
        read = read + [rng.randrange(4) for _ in range(5)] + cp
    n = len(read)
    if n == 0:
        return None
    kerr = max_unit_errors(T, m, match, mismatch, gopen, gext)
    if kerr is None or kerr < 1 or m + kerr > 48:
        return None
    c_max = max(match, match - mismatch, match + max(gopen, gext))
    T_lo = max(T, match * m - c_max)
    k_lo = max_unit_errors(T_lo, m, match, mismatch, gopen, gext)
    if k_lo is None or not (1 <= k_lo < kerr):
        return None
    sc = Scan(ad, read, T, kerr, match, mismatch, gopen, gext)
    D = sc.D
    det = [j for j in range(1, n + 1) if D[j] <= sc.kthr]
    ncl, c = clusters(det, m + kerr)
    own = 4 * (n // 4)
    dmin = min(D[1: own + 1]) if own else m
    if not (dmin <= 1 and ncl == 1):
        return None
    dl = sc.shrink(T_lo, k_lo)
    lo = c[0] + dl if c[0] > 1 else 1
    hi = c[1] - dl if c[1] < n else n
    if not (hi >= lo and hi - lo + 2 * k_lo + 1 <= KBAND):
        return None
    kband = KBAND_N if hi - lo + 2 * k_lo + 1 <= KBAND_N else KBAND       # (the narrow class: round 4, also from the Myers stage)
    av = lo - m - k_lo
    letters = b"ACGT"
    qb = bytes(letters[x] for x in ad)
    rb = bytes(letters[x] for x in read)
    fs, fl, _, _ = pyref.sg_stats_py(qb, rb, match, mismatch, gopen, gext)
    bs, bl = banded(ad, read, av, match, mismatch, gopen, gext, kband)
    ok_full, ok_band = fs >= T, bs >= T
    good = ok_full == ok_band and (not ok_full or (fs == bs and fl == bl))
    if not good:
        print("MISMATCH", dict(match=match, mismatch=mismatch, gopen=gopen, gext=gext, T=T, kerr=kerr, k_lo=k_lo, av=av, lo=lo, hi=hi,
                               cl=c, dmin=dmin), qb, rb, "full", (fs, fl), "band", (bs, bl))
    return good


# ---- the block filter (k_bfilter): class N / class 0 without a Myers pass --------------------------------------
KBAND_N = 5


def block_plan(m, K):
    """K + 1 disjoint adapter blocks (host: fill_dp): stretches of m // (K+1) bases (the first m % (K+1) one longer),
    the first min(stretch, 8) bases of each; None when a block would be shorter than 5 bases or K + 1 > 4"""
    nb = K + 1
    if K < 1 or nb > 4:
        return None
    base, rem = divmod(m, nb)
    offs, lens, o = [], [], 0
    for b in range(nb):
        ln = base + (1 if b < rem else 0)
        offs.append(o)
        lens.append(min(ln, 8))
        o += ln
    return (offs, lens) if min(lens) >= 5 else None


def block_filter(ad, read, K, plan, row_bases):
    """k_bfilter restated.  Returns None (task goes on to the Myers filter) or (band width class, av)."""
    m, n = len(ad), len(read)
    offs, lens = plan
    row = list(read) + [0] * (row_bases + 48 - n)          # the packed row is zero padded ('A')
    diags = []
    for o, ln in zip(offs, lens):
        for q in range(row_bases):                           # every start position of the row, as the kernel's masks hold them
            if row[q:q + ln] == ad[o:o + ln]:
                if q < o:
                    return None                              # a copy whose adapter would start left of the read: not handled here
                diags.append(q - o)
    if not diags:
        return None
    dmin, dmax = min(diags), max(diags)
    if dmax + m > n:
        return None                                          # a copy (or padding) whose adapter would run past the read end
    def ham(d):
        return sum(a != b for a, b in zip(ad, read[d:d + m]))
    if not (ham(dmin) <= 1 or (dmax != dmin and ham(dmax) <= 1)):
        return None                                          # no column with unit edit distance <= 1 certified
    width = dmax - dmin + 2 * K + 1
    if width <= KBAND_N:
        return KBAND_N, dmin - K
    if width <= KBAND:
        return KBAND, dmin - K
    return None


def block_case(rng, stats=None):
    """one random task through the block filter; None = not classified by it, else True/False = banded DP == full DP"""
    match = rng.choice([1, 2, 3, 3, 3, 4, 5])
    mismatch = rng.choice([-4, -3, -2, -2, -2, -1, 0])
    gopen = rng.choice([0, 2, 3, 5, 5, 5, 8])
    gext = rng.choice([1, 2, 2, 2, 3, 4])
    m = rng.randint(10, 32)
    acc = rng.choice([0.5, 0.6, 0.7, 0.75, 0.75, 0.8, 0.9])
    T = int(acc * match * m // 1) + 1
    ad = [rng.randrange(4) for _ in range(m)]
    if rng.random() < 0.15:                                  # repetitive adapters: blocks that look alike, homopolymer runs
        u = [rng.randrange(4) for _ in range(rng.randint(1, 4))]
        ad = (u * m)[:m]
        for _ in range(rng.randint(0, 3)):
            ad[rng.randrange(m)] = rng.randrange(4)
    cp = list(ad)
    for _ in range(rng.choice([1, 1, 1, 1, 1, 2, 2, 3])):
        t = rng.random()
        p = rng.randrange(len(cp))
        if t < 0.7:
            cp[p] = (cp[p] + rng.randint(1, 3)) % 4
        elif t < 0.85:
            cp.insert(p, rng.randrange(4))
        elif len(cp) > 2:
            del cp[p]
    flank = [rng.randrange(4) for _ in range(rng.choice([0, 0, 0, 1, 2, 3, 5, 9, 17, 40]))]
    tail = [rng.randrange(4) for _ in range(rng.choice([0, 0, 0, 1, 2, 3, 7, 20, 60]))]
    read = flank + cp + tail
    if rng.random() < 0.15:
        read = read[rng.randint(1, 3):]
    if rng.random() < 0.15:
        read = read[: len(read) - rng.randint(1, 3)]
    if rng.random() < 0.15:                                  # a second copy, close by or far away, intact or damaged
        cp2 = list(ad)
        for _ in range(rng.randint(0, 3)):
            cp2[rng.randrange(m)] = rng.randrange(4)
        read = read + [rng.randrange(4) for _ in range(rng.choice([0, 1, 2, 4, 9, 30]))] + cp2[rng.randint(0, 3):]
    if rng.random() < 0.1:                                   # a stray copy of one block
        o = rng.randrange(max(m - 7, 1))
        at = rng.randrange(len(read) + 1)
        read = read[:at] + ad[o:o + 7] + read[at:]
    n = len(read)
    if n == 0 or n > 640:
        return None
    kerr = max_unit_errors(T, m, match, mismatch, gopen, gext)
    if kerr is None or kerr < 1 or m + kerr > 48:
        return None
    c_max = max(match, match - mismatch, match + max(gopen, gext))
    T_lo = max(T, match * m - c_max)
    k_lo = max_unit_errors(T_lo, m, match, mismatch, gopen, gext)
    if k_lo is None or not (1 <= k_lo < kerr):
        return None
    plan = block_plan(m, k_lo)
    if plan is None:
        return None
    row_bases = 160 if n <= 160 else 320 if n <= 320 else 640
    cls = block_filter(ad, read, k_lo, plan, row_bases)
    if cls is None:
        return None
    kband, av = cls
    letters = b"ACGT"
    qb = bytes(letters[x] for x in ad)
    rb = bytes(letters[x] for x in read)
    if rb.find(qb) >= 0:
        return None                                          # an exact copy: the scan takes the read, no alignment task exists
    fs, fl, _, _ = pyref.sg_stats_py(qb, rb, match, mismatch, gopen, gext)
    bs, bl = banded(ad, read, av, match, mismatch, gopen, gext, kband)
    ok_full, ok_band = fs >= T, bs >= T
    good = ok_full == ok_band and (not ok_full or (fs == bs and fl == bl))
    if stats is not None:
        stats[kband] = stats.get(kband, 0) + 1
    if not good:
        print("BLOCK MISMATCH", dict(match=match, mismatch=mismatch, gopen=gopen, gext=gext, T=T, kerr=kerr, k_lo=k_lo, av=av, kband=kband, plan=plan),
              qb, rb, "full", (fs, fl), "band", (bs, bl))
    return good


# ---- the wide band (class 6 of the Myers filter): ANY single-cluster task whose cluster is narrow ------------------------
# Every accepting alignment ends in a column of the cluster [jmin, jmax] (unit edit distance <= kerr there) and has at most
# kerr unit edits, so all of its cells lie on the diagonals [jmin - m - kerr, jmax - m + kerr]: when those are at most
# KBAND_W the task is a banded one, whatever its edits are (no certified column of distance <= 1 is needed: the bound is the
# plain kerr, not the tightened one).  15 % substitutions put nearly every task here instead of into a 64-column window.
KBAND_W = 16
KHAM_DIAGS = 12


def wide_case(rng, stats=None):
    match = rng.choice([1, 2, 3, 3, 3, 4, 5])
    mismatch = rng.choice([-4, -3, -2, -2, -2, -1, 0])
    gopen = rng.choice([0, 2, 3, 5, 5, 5, 8])
    gext = rng.choice([1, 2, 2, 2, 3, 4])
    m = rng.randint(4, 32)
    acc = rng.choice([0.5, 0.6, 0.7, 0.75, 0.75, 0.8, 0.9])
    T = int(acc * match * m // 1) + 1
    ad = [rng.randrange(4) for _ in range(m)]
    cp = list(ad)
    for _ in range(rng.choice([0, 1, 2, 2, 3, 3, 4, 5, 6])):
        t = rng.random()
        p = rng.randrange(len(cp))
        if t < 0.75:
            cp[p] = (cp[p] + rng.randint(1, 3)) % 4
        elif t < 0.88:
            cp.insert(p, rng.randrange(4))
        elif len(cp) > 2:
            del cp[p]
    flank = [rng.randrange(4) for _ in range(rng.choice([0, 0, 1, 2, 5, 9, 17, 40]))]
    tail = [rng.randrange(4) for _ in range(rng.choice([0, 0, 1, 2, 3, 7, 20, 60]))]
    read = flank + cp + tail
    if rng.random() < 0.25:
        read = read[rng.randint(1, 4):]
    if rng.random() < 0.25:
        read = read[: len(read) - rng.randint(1, 4)]
    n = len(read)
    if n == 0:
        return None
    kerr = max_unit_errors(T, m, match, mismatch, gopen, gext)
    if kerr is None or kerr < 1 or m + kerr > 48:
        return None
    sc = Scan(ad, read, T, kerr, match, mismatch, gopen, gext)
    D = sc.D
    det = [j for j in range(1, n + 1) if D[j] <= sc.kthr]
    if not det:
        return None
    ncl, c = clusters(det, m + kerr)
    if ncl != 1 or (c[1] - c[0]) + 2 * kerr + 1 > KBAND_W:
        return None
    av = c[0] - m - kerr
    letters = b"ACGT"
    qb = bytes(letters[x] for x in ad)
    rb = bytes(letters[x] for x in read)
    fs, fl, _, _ = pyref.sg_stats_py(qb, rb, match, mismatch, gopen, gext)
    bs, bl = banded(ad, read, av, match, mismatch, gopen, gext, KBAND=KBAND_W)
    ok_full, ok_band = fs >= T, bs >= T
    good = ok_full == ok_band and (not ok_full or (fs == bs and fl == bl))
    if stats is not None:
        stats["accepted" if ok_full else "rejected"] = stats.get("accepted" if ok_full else "rejected", 0) + 1
    if not good:
        print("MISMATCH (wide band)", dict(match=match, mismatch=mismatch, gopen=gopen, gext=gext, T=T, kerr=kerr, av=av, cl=c), qb, rb,
              "full", (fs, fl), "band", (bs, bl))
    return good


# ---- the Hamming certificate (round 4): banded classes for single-cluster tasks with two or more substitutions ---------------
# The adapter laid on the cluster's centre diagonal(s) differs from the read in h bases: an ungapped alignment of score
# match*m - h*(match - mismatch), hence S* >= T'' = max(T, that) and every optimal alignment has at most
# kq = max_unit_errors(T'') unit edits.  End columns of optimal alignments lie kerr - kq inside the cluster, their cells on
# [lo - m - kq, hi - m + kq] (vf_k_front.hip, hamming_band_class / myers_classify).
def hamming_decision(ad, read, T, kerr, match, mismatch, gopen, gext, k_lo=None, kq_shift=0):
    """the Myers pass's class decision for one task as myers_classify makes it, restricted to what the certificate adds:
    returns None (the certificate does not apply / the task is class 0 by the d <= 1 rule already) or (band width, av, h, kq)"""
    m, n = len(ad), len(read)
    sc = Scan(ad, read, T, kerr, match, mismatch, gopen, gext)
    D = sc.D
    det = [j for j in range(1, n + 1) if D[j] <= sc.kthr]
    if not det:
        return None
    ncl, c = clusters(det, m + kerr)
    if ncl != 1:
        return None
    jmin, jmax = c[0], c[1]
    own = 4 * (n // 4)
    dmin = min(D[1: own + 1]) if own else m
    if k_lo is not None and dmin <= 1:                       # class 0 by the tightened bound of a d <= 1 column (one_case covers it)
        c_max = max(match, match - mismatch, match + max(gopen, gext))
        dl = sc.shrink(max(T, match * m - c_max), k_lo)
        lo = jmin + dl if jmin > 1 else 1
        hi = jmax - dl if jmax < n else n
        if hi >= lo and hi - lo + 2 * k_lo + 1 <= KBAND:
            return None
    hmin = None
    j0, cnt = jmin, jmax - jmin + 1
    if cnt > KHAM_DIAGS:                                     # the centre-most diagonals of a wide cluster
        j0 += (cnt - KHAM_DIAGS) >> 1
        cnt = KHAM_DIAGS
    for jc in range(j0, j0 + cnt):
        d = jc - m
        if d < 0 or jc > n or d - max(j0 - m, 0) >= KHAM_DIAGS:
            continue
        h = sum(a != b for a, b in zip(ad, read[d:d + m]))
        hmin = h if hmin is None else min(hmin, h)
    if hmin is None or not (1 <= hmin <= 7) or hmin > kerr:
        return None
    T2 = max(T, match * m - hmin * (match - mismatch))
    kq = max_unit_errors(T2, m, match, mismatch, gopen, gext)
    if kq is None or not (1 <= kq < kerr):
        return None
    kq_used = kq - kq_shift                                  # (kq_shift > 0: a deliberately wrong bound, to see the model bite)
    if kq_used < 1:
        return None
    dl = sc.shrink(T2, kq) + kq_shift
    lo = jmin + dl if jmin > 1 else 1
    hi = jmax - dl if jmax < n else n
    if hi < lo:
        return None
    width = hi - lo + 2 * kq_used + 1
    if width <= KBAND_N:
        return KBAND_N, lo - m - kq_used, hmin, kq
    if width <= KBAND:
        return KBAND, lo - m - kq_used, hmin, kq
    if width <= KBAND_W:
        return KBAND_W, lo - m - kq_used, hmin, kq
    return None


def hamming_case(rng, stats=None, kq_shift=0):
    """one random task; None = the certificate does not classify it, else True/False = banded DP == full-matrix DP"""
    match = rng.choice([1, 2, 3, 3, 3, 4, 5])
    mismatch = rng.choice([-4, -3, -2, -2, -2, -1, 0])
    gopen = rng.choice([0, 2, 3, 5, 5, 5, 8])
    gext = rng.choice([1, 2, 2, 2, 3, 4])
    m = rng.randint(8, 32)
    acc = rng.choice([0.5, 0.6, 0.7, 0.75, 0.75, 0.75, 0.8])
    T = int(acc * match * m // 1) + 1
    ad = [rng.randrange(4) for _ in range(m)]
    if rng.random() < 0.1:                                   # repetitive adapters: shifted diagonals look alike
        u = [rng.randrange(4) for _ in range(rng.randint(1, 4))]
        ad = (u * m)[:m]
        for _ in range(rng.randint(0, 3)):
            ad[rng.randrange(m)] = rng.randrange(4)
    cp = list(ad)
    for _ in range(rng.choice([2, 2, 2, 2, 3, 3, 4, 5])):
        t = rng.random()
        p = rng.randrange(len(cp))
        if t < 0.82:
            cp[p] = (cp[p] + rng.randint(1, 3)) % 4
        elif t < 0.91:
            cp.insert(p, rng.randrange(4))
        elif len(cp) > 2:
            del cp[p]
    flank = [rng.randrange(4) for _ in range(rng.choice([0, 0, 1, 2, 3, 5, 9, 17, 40]))]
    tail = [rng.randrange(4) for _ in range(rng.choice([0, 0, 1, 2, 3, 7, 20, 60]))]
    read = flank + cp + tail
    if rng.random() < 0.2:
        read = read[rng.randint(1, 4):]
    if rng.random() < 0.2:
        read = read[: len(read) - rng.randint(1, 4)]
    if rng.random() < 0.08:                                  # a second, damaged copy close by: one cluster, two candidates
        cp2 = list(ad)
        for _ in range(rng.randint(1, 4)):
            cp2[rng.randrange(m)] = rng.randrange(4)
        read = read + [rng.randrange(4) for _ in range(rng.choice([0, 1, 2, 4]))] + cp2
    n = len(read)
    if n == 0 or n > 640:
        return None
    kerr = max_unit_errors(T, m, match, mismatch, gopen, gext)
    if kerr is None or kerr < 2 or m + kerr > 48:
        return None
    c_max = max(match, match - mismatch, match + max(gopen, gext))
    k_lo = max_unit_errors(max(T, match * m - c_max), m, match, mismatch, gopen, gext)
    if k_lo is None or not (1 <= k_lo < kerr):
        k_lo = None
    got = hamming_decision(ad, read, T, kerr, match, mismatch, gopen, gext, k_lo, kq_shift)
    if got is None:
        return None
    kband, av, h, kq = got
    letters = b"ACGT"
    qb = bytes(letters[x] for x in ad)
    rb = bytes(letters[x] for x in read)
    if rb.find(qb) >= 0:
        return None                                          # an exact copy: no alignment task exists
    fs, fl, _, _ = pyref.sg_stats_py(qb, rb, match, mismatch, gopen, gext)
    bs, bl = banded(ad, read, av, match, mismatch, gopen, gext, kband)
    ok_full, ok_band = fs >= T, bs >= T
    good = ok_full == ok_band and (not ok_full or (fs == bs and fl == bl))
    if stats is not None:
        stats[(kband, h)] = stats.get((kband, h), 0) + 1
        stats["accepted" if ok_full else "rejected"] = stats.get("accepted" if ok_full else "rejected", 0) + 1
    if not good:
        print("MISMATCH (Hamming certificate)", dict(match=match, mismatch=mismatch, gopen=gopen, gext=gext, T=T, kerr=kerr, h=h, kq=kq, av=av, kband=kband),
              qb, rb, "full", (fs, fl), "band", (bs, bl))
    return good


# ---- the indel scan (round 4): the filter itself ---------------------------------------------------------------------------------
def scan_case(rng, stats=None, units_shift=0):
    """one random task under a scheme for which the host picks the indel scan: True/False = (the kernel's two-lattice scan equals
    the plain recurrence) and (an alignment the full-matrix DP accepts ends in a column the scan detects: C <= kthr there);
    None = the scheme keeps the unit scan.  units_shift > 0: a threshold that much too small, to see the check bite"""
    match = rng.choice([1, 2, 3, 3, 3, 4, 5])
    mismatch = rng.choice([-4, -3, -2, -2, -2, -1, 0])
    gopen = rng.choice([1, 2, 3, 5, 5, 5, 8])
    gext = rng.choice([1, 2, 2, 2, 3, 4])
    m = rng.randint(4, 32)
    acc = rng.choice([0.5, 0.6, 0.7, 0.75, 0.75, 0.8, 0.9])
    T = int(acc * match * m // 1) + 1
    ad = [rng.randrange(4) for _ in range(m)]
    cp = list(ad)
    for _ in range(rng.choice([0, 1, 1, 2, 2, 3, 3, 4, 5])):
        t = rng.random()
        p = rng.randrange(len(cp))
        if t < 0.5:
            cp[p] = (cp[p] + rng.randint(1, 3)) % 4
        elif t < 0.8:
            for _ in range(rng.choice([1, 1, 2, 3, 5])):
                cp.insert(p, rng.randrange(4))
        elif len(cp) > 3:
            del cp[p:p + rng.choice([1, 1, 2])]
    flank = [rng.randrange(4) for _ in range(rng.choice([0, 0, 1, 2, 5, 9, 17, 40]))]
    tail = [rng.randrange(4) for _ in range(rng.choice([0, 0, 1, 2, 3, 7, 20, 60]))]
    read = flank + cp + tail
    if rng.random() < 0.3:
        read = read[rng.randint(1, 6):]
    if rng.random() < 0.3:
        read = read[: len(read) - rng.randint(1, 6)]
    n = len(read)
    if n == 0:
        return None
    kerr = max_unit_errors(T, m, match, mismatch, gopen, gext)
    if kerr is None or kerr < 1 or m + kerr > 48:
        return None
    sc = Scan(ad, read, T, kerr, match, mismatch, gopen, gext)
    if not sc.indel:
        return None
    good = indel_last_row_bitparallel(ad, read) == sc.D
    letters = b"ACGT"
    qb = bytes(letters[x] for x in ad)
    rb = bytes(letters[x] for x in read)
    fs, fl, bq, bd = pyref.sg_stats_py(qb, rb, match, mismatch, gopen, gext)
    if stats is not None:
        stats["accepted" if fs >= T else "rejected"] = stats.get("accepted" if fs >= T else "rejected", 0) + 1
        if fs < T and min(sc.D[1:]) > sc.kthr >= min(myers_last_row(ad, read)[1:]) - 0 and min(myers_last_row(ad, read)[1:]) <= kerr:
            stats["turned away (the unit scan would have passed it on)"] = stats.get("turned away (the unit scan would have passed it on)", 0) + 1
    if fs >= T and sc.D[bd] > sc.kthr - units_shift:
        good = False
    if not good:
        print("MISMATCH (indel scan)", dict(match=match, mismatch=mismatch, gopen=gopen, gext=gext, T=T, kerr=kerr, kthr=sc.kthr), qb, rb,
              "full", (fs, fl, bq, bd), "C", sc.D[bd])
    return good


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ran = bad = 0
    while ran < cases:
        g = one_case(rng)
        if g is None:
            continue
        ran += 1
        bad += not g
    print(f"{ran} class-0 tasks, {bad} mismatches")
    ranb = badb = 0
    stats = {}
    while ranb < cases:
        g = block_case(rng, stats)
        if g is None:
            continue
        ranb += 1
        badb += not g
    print(f"{ranb} block-filter tasks ({stats}), {badb} mismatches")
    ranh = badh = 0
    hstats = {}
    while ranh < cases:
        g = hamming_case(rng, hstats)
        if g is None:
            continue
        ranh += 1
        badh += not g
    print(f"{ranh} Hamming-certified tasks ({dict(sorted((str(k), v) for k, v in hstats.items()))}), {badh} mismatches")
    return 1 if bad or badb or badh else 0


if __name__ == "__main__":
    sys.exit(main())
