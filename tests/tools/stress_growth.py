"""stress: parallel FASTQ reader (16 threads) + table growth from a tiny table under concurrent submitters"""
import os, sys, time, gzip
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vfind_amd
from benchlib import synth
from oracle import oracle
ads = synth.adapters(0x5EED0003, 20)
gen = synth.Generator(0x5EED0003, ads)
path = "/tmp/vf_stress.fq.gz"
n, members = 6_000_000, 24
if not os.path.exists(path):
    per = n // members
    q = b"F" * 150
    with open(path, "wb") as f:
        for m in range(members):
            b, o = gen.ascii(m * per, per, 16)
            seqs = b.reshape(-1, 150)
            f.write(gzip.compress(b"".join(b"@r\n" + seqs[i].tobytes() + b"\n+\n" + q + b"\n" for i in range(per)), 1))
want = oracle.count_fastq(path, ads, n_threads=16)
for slots, br in ((1 << 12, 1 << 14), (1 << 10, 1 << 12)):
    eng = vfind_amd.Engine(ads, n_threads=16, queue_len=6, table_slots=slots, batch_reads=br)
    t0 = time.perf_counter(); eng.run_fastq(path); dt = time.perf_counter() - t0
    got = eng.finish_dict(); st = eng.stats()
    print(f"initial slots {slots}, batch {br}: {dt:.2f}s, entries {st['table_entries']}, equal to oracle: {got == want}", flush=True)
    eng.close()
