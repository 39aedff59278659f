"""per-wave timeline of k_finish (diagnosis build, `make -C vfind_amd/csrc ablation`): where a cold launch spends its time -
first trips, the steady fast path, the insert queue.  usage: fin_trace.py [keep]   (keep = warm table: no clear between passes)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfind_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libvfind_hip_ablation.so")
import numpy as np, torch
import vfind_amd
from benchlib import synth
keep = len(sys.argv) > 1 and sys.argv[1] == "keep"
n = 10_000_000
ads = synth.adapters(0x5EED0002, 20)
gen = synth.Generator(0x5EED0002, ads)
rows_h, lens_h = gen.packed(0, n, 16)
rows = torch.from_numpy(rows_h).cuda(); lens = torch.from_numpy(lens_h.view(np.int16)).cuda()
eng = vfind_amd.Engine(ads, table_slots=1 << 21)
L = _lib.lib()
for it in range(3):
    if not keep or it == 0:
        eng.clear()
    eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n)
    eng.sync()
st = eng.stats()
print("insert_path reads (3 passes):", st["insert_path"], "counted", st["counted"], "entries", st["table_entries"])
tr = np.zeros((2048 * 4, 12), np.uint64)
L.vf_debug_fin_trace.argtypes = [C.c_void_p]
assert L.vf_debug_fin_trace(tr.ctypes.data) == 0
t = tr[tr[:, 0] > 0].astype(np.int64)
z = t[:, 0].min()
us = lambda a: (a - z) / 100.0
print(f"{len(t)} waves; starts over {us(t[:,0]).max():.1f} us; kernel span {us(t[:,9]).max():.1f} us")
names = ["start", "trip0", "trip1", "trip2", "trip4", "trip8", "trip16", "trip32", "last trip", "end"]
for i, nm in enumerate(names):
    v = us(t[:, i])
    print(f"  {nm:10s} mean {v.mean():8.1f}  p10 {np.percentile(v,10):8.1f}  p90 {np.percentile(v,90):8.1f}  max {v.max():8.1f} us")
print(f"  insert-queue batches per wave: mean {t[:,10].mean():.2f} max {t[:,10].max()}; time in them per wave: mean {t[:,11].mean()/100:.1f} us, max {t[:,11].max()/100:.1f} us")
