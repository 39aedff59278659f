"""the FASTQ path (device framing, several slots in flight, table growth) run again and again on one file: every run must
give the oracle's table and the counts must sum to the reads the kernels say they counted.  Prints what differs."""
import gzip, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vfind_amd
from oracle import oracle
from tests import synth

AD = ("ACGTTGCAAGCTTGCATGCA", "TTGACCGTAAGCTAGGCTAA")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 16
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 16
os.environ.setdefault("VFIND_INFLATE_PARALLEL", "1")
os.environ.setdefault("VFIND_PINFLATE_CELL", "262144")
reads = synth.make_reads(84, 120000, AD, L=150, sub=0.01, indel=0.002, n_variants=800)
want = oracle.count(reads, AD)
p = os.path.join(tempfile.mkdtemp(), "single.fq.gz")
open(p, "wb").write(gzip.compress(synth.fastq_text(reads), 1))
bad = 0
for it in range(runs):
    with vfind_amd.Engine(AD, n_threads=nt, table_slots=slots) as e:
        e.run_fastq(p)
        got = e.finish_dict()
        st = e.stats()
    tot = sum(got.values())
    if got != want or tot != st["counted"]:
        bad += 1
        diff = {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)}
        print(f"run {it}: sum {tot} counted {st['counted']} oracle {sum(want.values())} entries {st['table_entries']} differing {len(diff)}: {sorted(diff.items())[:6]}", flush=True)
print(f"{runs} runs, {nt} threads, {slots} slots, VFIND_COUNT_LOG={os.environ.get('VFIND_COUNT_LOG', '-')}: {bad} differ", flush=True)
sys.exit(1 if bad else 0)
