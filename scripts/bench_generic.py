import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vfind_amd
from benchlib import synth as vsynth
ads = vsynth.adapters(0x5EED0002, 20)
gen = vsynth.Generator(0x5EED0002, ads, n_variants=100000)
n = 2_000_000
data, offs = gen.ascii(0, n, 16)
for frac in (0.0, 0.001, 0.01, 0.05):
    d = data.copy()
    rng = np.random.default_rng(1)
    k = int(n * frac)
    if k:
        idx = rng.choice(n, k, replace=False)
        pos = rng.integers(0, 150, k)
        d[idx * 150 + pos] = ord('N')
    eng = vfind_amd.Engine(ads, n_threads=16, queue_len=4, table_slots=1 << 21)
    eng.submit((d, offs)); eng.sync(); eng.clear()
    t0 = time.perf_counter(); eng.submit((d, offs)); eng.sync(); dt = time.perf_counter() - t0
    st = eng.stats()
    print(f"N-fraction {frac}: {dt:.3f}s {n/dt/1e6:.2f} M reads/s generic={st['reads_generic']} ms_generic={st['ms_generic']:.2f}", flush=True)
    eng.close()
