"""End-to-end timing of vf_run_fastq (gzip FASTQ -> table): BASELINE configs[2]-style input at a
reduced read count.  Writes a multi-member gzip FASTQ of synthetic reads, then times the whole call."""
import argparse, gzip, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import vfind_amd
from benchlib import synth

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=4_000_000)
ap.add_argument("--members", type=int, default=16)
ap.add_argument("--out", default="/tmp/vf_bench.fq.gz")
ap.add_argument("--threads", type=int, nargs="*", default=[1, 3, 8, 16])
ap.add_argument("--check", action="store_true")
ap.add_argument("--level", type=int, default=1, help="gzip level of the generated file")
ap.add_argument("--random-quals", action="store_true", help="quality strings with Illumina-like binned scores instead of all F")
a = ap.parse_args()

ads = synth.adapters(0x5EED0003, 20)
gen = synth.Generator(0x5EED0003, ads)
if not os.path.exists(a.out) or os.path.getsize(a.out) == 0:
    t0 = time.time()
    per = (a.reads + a.members - 1) // a.members
    qual = b"F" * 150
    qrng = np.random.default_rng(99)
    with open(a.out, "wb") as f:
        for m in range(a.members):
            lo, hi = m * per, min(a.reads, (m + 1) * per)
            if lo >= hi:
                break
            b, o = gen.ascii(lo, hi - lo, 16)
            seqs = b.reshape(-1, 150)
            parts = []
            for i in range(hi - lo):
                parts.append(b"@r%d\n" % (lo + i)); parts.append(seqs[i].tobytes()); parts.append(b"\n+\n")
                parts.append(qrng.choice(np.frombuffer(b"FFFFFFFFFFFF::,#", np.uint8), 150).tobytes() if a.random_quals else qual); parts.append(b"\n")
            f.write(gzip.compress(b"".join(parts), compresslevel=a.level))
    print(f"wrote {a.out}: {os.path.getsize(a.out)/1e6:.1f} MB gz, {a.reads} reads, {a.members} members in {time.time()-t0:.1f}s", flush=True)
text_bytes = None
for nt in a.threads:
    eng = vfind_amd.Engine(ads, n_threads=nt, queue_len=4, table_slots=1 << 22)
    for run in ("first call (allocates staging and workspaces)", "second call"):
        eng.clear(); eng.reset_stats()
        t0 = time.perf_counter()
        eng.run_fastq(a.out)
        dt = time.perf_counter() - t0
        st = eng.stats()
        print(f"n_threads={nt} {run}: {dt:.2f}s  {st['reads']/dt/1e6:.2f} M reads/s  entries={st['table_entries']} counted={st['counted']}", flush=True)
    if a.check and nt == a.threads[-1]:
        from oracle import oracle
        got = eng.finish_dict()
        want = oracle.count_fastq(a.out, ads, n_threads=16)
        print("parity with oracle:", got == want, len(got))
    eng.close()
