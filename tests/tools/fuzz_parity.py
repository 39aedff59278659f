"""differential fuzz: random adapters (lengths 1-32, different for the two ends), scoring schemes (including ones that
switch the Myers filter off or leave the 2-bit path), thresholds, read lengths and error rates; per-read spans and the
(sequence,count) table from the device path must equal the CPU oracle's, and the result build must have folded no row
(vf_stats.dup_rows: one context, one table - a duplicate entry is a protocol error even though its counts are summed).
usage: fuzz_parity.py [iterations] [seed]        (tests/test_gpu_parity.py::test_fuzz_parity_sample runs `run` in the suite)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def run(iters, seed, n_reads=1500, verbose=True, long_reads=False):
    """-> list of (iteration, description) of every configuration whose device results differ from the oracle's"""
    import vfind_amd
    from oracle import oracle
    from tests import synth
    rng = np.random.default_rng(seed)
    bad = []
    for it in range(iters):
        mp, ms = int(rng.integers(1, 33)), int(rng.integers(1, 33))
        ads = ("".join(rng.choice(list("ACGT"), mp)), "".join(rng.choice(list("ACGT"), ms)))
        kw = dict(match_score=int(rng.integers(1, 6)), mismatch_score=int(rng.integers(-4, 1)),
                  gap_open_penalty=int(rng.integers(0, 9)), gap_extend_penalty=int(rng.integers(0, 5)),
                  accept_prefix_alignment=float(rng.choice([0.5, 0.6, 0.75, 0.9, 1.0, 0.83])),
                  accept_suffix_alignment=float(rng.choice([0.5, 0.6, 0.75, 0.9, 1.0, 0.67])))
        L = int(rng.choice([150, 300, 321, 420, 500, 639, 640] if long_reads else [40, 100, 150, 151, 250, 300]))     # (long: the 20- and 40-word rows)
        if L < mp + ms + 12:
            L = mp + ms + 40
        reads = synth.make_reads(int(rng.integers(1, 1 << 30)), n_reads, ads, L=L, sub=float(rng.choice([0.0, 0.02, 0.1])),
                                 indel=float(rng.choice([0.0, 0.01, 0.05])), vary_len=bool(rng.integers(0, 2)),
                                 dirty=float(rng.choice([0.0, 0.02])), n_variants=int(rng.choice([5, 200])))
        skip = bool(rng.integers(0, 2))
        why = ""
        try:
            with vfind_amd.Engine(ads, table_slots=1 << 14, skip_translation=skip, **kw) as e:
                st, en = e.spans(reads)
                e.submit(reads)
                got = e.finish_dict()
                dup = e.stats()["dup_rows"]
            ost, oen = oracle.spans(reads, ads, **kw)
            want = oracle.count(reads, ads, skip_translation=skip, **kw)
            if not ((st == ost).all() and (en == oen).all()):
                why = "spans"
            elif got != want:
                why = "table"
            elif dup:
                why = f"{dup} duplicate rows folded"
        except Exception as ex:                               # noqa: BLE001
            why = f"exception {type(ex).__name__}: {ex}"
        if why:
            bad.append((it, f"{why}: adapters {ads} {kw} L={L} skip_translation={skip}"))
            if verbose:
                print("MISMATCH", it, bad[-1][1], flush=True)
    return bad


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    long_reads = len(sys.argv) > 3 and sys.argv[3] == "long"          # reads of 150-640 bp: the 20- and 40-word kernels
    bad = run(iters, seed, n_reads=600 if long_reads else 1500, long_reads=long_reads)
    print(f"{iters} iterations{' (long reads)' if long_reads else ''}, {len(bad)} mismatches, {time.time() - t0:.1f}s", flush=True)
    sys.exit(1 if bad else 0)
