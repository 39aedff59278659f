"""host-side time of every vf_batch_acquire / vf_batch_submit of the pinned batch API leg (who blocks the submitting thread?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vfind_amd
from benchlib import synth
ads = synth.adapters(0x5EED0002, 20)
gen = synth.Generator(0x5EED0002, ads)
W = gen.row_dwords
ql, nb = int(os.environ.get("QL", "8")), int(os.environ.get("NB", "1250000"))
eng = vfind_amd.Engine(ads, queue_len=ql, table_slots=int(os.environ.get("SLOTS", str(1 << 21))))
for k in range(ql):
    h, r, l = eng.batch_acquire(nb, W)
    gen.packed_into(k * nb, nb, r.ctypes.data, l.ctypes.data, 16)
    eng.batch_submit(h, nb)
eng.sync()
for _ in range(2 * ql):                                   # untimed, at full submit rate: the table grows for a whole pipeline once
    h, r, l = eng.batch_acquire(nb, W)
    eng.batch_submit(h, nb)
eng.sync(); eng.clear()
ta, ts = [], []
t0 = time.perf_counter()
for _ in range(4 * ql):
    a = time.perf_counter()
    h, r, l = eng.batch_acquire(nb, W)
    b = time.perf_counter()
    eng.batch_submit(h, nb)
    c = time.perf_counter()
    ta.append((b - a) * 1e3); ts.append((c - b) * 1e3)
eng.sync()
dt = time.perf_counter() - t0
print("acquire ms:", " ".join(f"{x:.2f}" for x in ta))
print("submit  ms:", " ".join(f"{x:.2f}" for x in ts))
print(f"{4 * ql * nb / dt / 1e6:.1f} M reads/s, {4 * ql * nb * (W * 4 + 2) / dt / 1e9:.2f} GB/s; table entries {eng.stats()['table_entries']}")
eng.close()
