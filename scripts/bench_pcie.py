"""PCIe-inclusive rate: packed reads in ordinary host memory -> vf_submit_packed (copy into pinned staging,
hipMemcpyAsync, kernels overlapped over queue_len slots) -> table.  Never the headline `value`."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vfind_amd
from benchlib import synth
n = 10_000_000
ads = synth.adapters(0x5EED0002, 20)
gen = synth.Generator(0x5EED0002, ads)
rows, lens = gen.packed(0, n, 16)
for ql, br, nt in ((2, 1 << 18, 3), (4, 1 << 20, 3), (4, 1 << 20, 8), (4, 1 << 20, 16)):
    eng = vfind_amd.Engine(ads, queue_len=ql, batch_reads=br, n_threads=nt, table_slots=1 << 22)
    eng.submit_packed(rows[:1 << 20], lens[:1 << 20]); eng.sync(); eng.clear()
    t0 = time.perf_counter()
    eng.submit_packed(rows, lens)
    eng.sync()
    dt = time.perf_counter() - t0
    print(f"queue_len={ql} batch_reads={br} n_threads={nt}: {n/dt/1e6:.1f} M reads/s ({n*42/dt/1e9:.2f} GB/s algorithmic, {n*40/dt/1e9:.1f} GB/s over PCIe)", flush=True)
    eng.close()
