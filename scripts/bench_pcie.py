"""PCIe-inclusive rate through the pinned batch API (vf_batch_acquire / vf_batch_submit): the staging of every slot is
filled once (the generator is far slower than the link) and re-submitted; variants of queue_len and batch size.
Never the headline `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vfind_amd
from benchlib import synth
n = 10_000_000
ads = synth.adapters(0x5EED0002, 20)
gen = synth.Generator(0x5EED0002, ads)
for ql, nb in ((2, 5_000_000), (4, 2_500_000), (4, 1_000_000), (8, 1_250_000), (8, 500_000), (3, 3_000_000)):
    eng = vfind_amd.Engine(ads, queue_len=ql, table_slots=1 << 21)
    W = gen.row_dwords
    for k in range(ql):
        h, r, l = eng.batch_acquire(nb, W)
        gen.packed_into(k * nb, nb, r.ctypes.data, l.ctypes.data, 16)
        eng.batch_submit(h, nb)
    eng.sync(); eng.clear()
    rounds = max(3, 40_000_000 // (ql * nb))
    t0 = time.perf_counter()
    for _ in range(rounds * ql):
        h, r, l = eng.batch_acquire(nb, W)
        eng.batch_submit(h, nb)
    eng.sync()
    dt = time.perf_counter() - t0
    reads = rounds * ql * nb
    print(f"queue_len={ql} batch={nb}: {reads/dt/1e6:.1f} M reads/s, {reads*(W*4+2)/dt/1e9:.2f} GB/s over the link", flush=True)
    eng.close()
