"""the FASTQ path alone, for a trace: writes (or reuses) a multi-member gzip FASTQ of the bench's read model and runs
vf_run_fastq + vf_finish on it.   usage: fastq_diag.py <reads> <n_threads> [path]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vfind_amd
from benchlib import synth, fastq
n, nt = int(sys.argv[1]), int(sys.argv[2])
path = sys.argv[3] if len(sys.argv) > 3 else "/tmp/vfind_diag.fq.gz"
seed = 0x5EED0002
ads = synth.adapters(seed, 20)
gen = synth.Generator(seed, ads, read_len=150, n_variants=1_000_000, sub_rate=0.005, indel_rate=0.0005)
if not os.path.exists(path):
    fastq.write_fastq_gz(path, gen, n, members=max(8, n // 160_000), threads=16)
with vfind_amd.Engine(ads, table_slots=1 << 21, arena_bytes=1 << 30, n_threads=nt, queue_len=4) as e:
    e.run_fastq(path)                       # allocations
    e.clear(); e.reset_stats()
    t0 = time.perf_counter()
    e.run_fastq(path)
    rows = len(e.finish_arrays()[2])
    dt = time.perf_counter() - t0
    st = e.stats()
print(f"fastq_diag: {n} reads, n_threads {nt}: {n / dt / 1e6:.2f} M reads/s, {dt:.3f} s, {rows} rows; chunks {st['fq_chunks']}, "
      f"inflate {st['fq_inflate_ms']:.0f} ms over {st['fq_inflate_threads']} threads, submit {st['fq_submit_ms']:.0f} ms (slot wait {st['fq_slot_wait_ms']:.0f}, "
      f"index wait {st['fq_index_wait_ms']:.0f}), sequencer waits: text {st['fq_seq_wait_ms']:.0f} ms, queue {st['fq_queue_full_ms']:.0f} ms", flush=True)
