"""one of the other BASELINE shapes, resident in HBM: M reads/s of a library build (A/B of DP changes)
usage: cfg_shape.py cfg4|cfg5 [lib.so] [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[2] != "-":
    from vfind_amd import _lib
    _lib.SO_PATH = os.path.abspath(sys.argv[2])
import numpy as np, torch
import vfind_amd
from benchlib import synth
shape = sys.argv[1]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8_000_000
L, m, V, sub, ind, seed = {"cfg4": (150, 20, 1_000_000, 0.15, 0.0, 0x5EED0004), "cfg5": (300, 30, 10_000_000, 0.005, 0.0005, 0x5EED0005),
                           "cfg2": (150, 20, 1_000_000, 0.005, 0.0005, 0x5EED0002)}[shape]
ads = synth.adapters(seed, m)
gen = synth.Generator(seed, ads, read_len=L, n_variants=V, sub_rate=sub, indel_rate=ind)
rows_h, lens_h = gen.packed(0, n, 16)
rows = torch.from_numpy(rows_h).cuda(); lens = torch.from_numpy(lens_h.view(np.int16)).cuda()
slots = 1 << 16
while slots < max(2 * min(V, n), n // 8):
    slots <<= 1
eng = vfind_amd.Engine(ads, table_slots=slots, arena_bytes=1 << 30)
def step():
    eng.clear(); eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n)
for _ in range(int(os.environ.get("CFG_WARM", "3"))): step()
eng.sync()
_w = eng.stats()
if any(_w.get(f"dbg{k}", 0) for k in range(4)): print("bounds check (warm-up):", [hex(_w[f"dbg{k}"]) for k in range(4)], flush=True)
eng.reset_stats(); eng.set_profiling(int(os.environ.get("CFG_PROF", "1")))
t0 = time.perf_counter()
for _ in range(10): step()
eng.sync()
dt = time.perf_counter() - t0
st = eng.stats()
if any(st.get(f"dbg{k}", 0) for k in range(4)):       # a bounds-check build (make variant NAME=bc DEFS=-DVF_BOUNDS_CHECK) reported
    print("bounds check:", [hex(st[f"dbg{k}"]) for k in range(4)], flush=True)
print(f"{shape} {os.path.basename(sys.argv[2]) if len(sys.argv) > 2 else 'default'}: {n * 10 / dt / 1e6:.1f} M reads/s, {dt / 10 * 1e3:.3f} ms/step; per step ms: scan {st['ms_scan']/10:.3f} filter {st['ms_filter']/10:.3f} dp {st['ms_dp_window']/10:.3f} finish {st['ms_finish']/10:.3f}", flush=True)
