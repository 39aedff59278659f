"""per-workgroup timeline of k_dpall (diagnosis build): class, start, prologue end, end of every workgroup, one batch lane"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfind_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libvfind_hip_ablation.so")
import numpy as np, torch
import vfind_amd
from benchlib import synth
n = 10_000_000
ads = synth.adapters(0x5EED0002, 20)
gen = synth.Generator(0x5EED0002, ads)
rows_h, lens_h = gen.packed(0, n, 16)
rows = torch.from_numpy(rows_h).cuda(); lens = torch.from_numpy(lens_h.view(np.int16)).cuda()
eng = vfind_amd.Engine(ads, table_slots=1 << 21)
L = _lib.lib()
for it in range(3):
    eng.clear()
    eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n)
    eng.sync()
tr = np.zeros((2, 2048, 4), np.uint64)
L.vf_debug_dp_trace.argtypes = [C.c_void_p]
assert L.vf_debug_dp_trace(tr.ctypes.data) == 0
for role, name in ((0, "prefix"), (1, "suffix")):
    t = tr[role][:1024].astype(np.int64)
    cls, t1, t2, t3 = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    live = t3 > 0
    z = t1[t1 > 0].min()
    print(f"{name}: workgroups started over {(t1[t1>0].max()-z)/100:.1f} us; kernel span {(max(t3.max(), t2.max())-z)/100:.1f} us")
    for c in (5, 0, 1, 2, 3, 4, 99):
        m = cls == c
        if not m.any():
            continue
        e = np.where(t3[m] > 0, t3[m], t2[m])
        print(f"  class {c:2d}: {m.sum():4d} workgroups, start {((t1[m]-z)/100).min():6.1f}..{((t1[m]-z)/100).max():6.1f} us, "
              f"prologue {((t2[m]-t1[m])/100).mean():5.1f} us, body mean {((e-t2[m])/100).mean():6.1f} max {((e-t2[m])/100).max():6.1f} us, last end {((e-z)/100).max():6.1f} us")
