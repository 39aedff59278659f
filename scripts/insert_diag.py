"""insert-path diagnosis (diagnosis build: make -C vfind_amd/csrc ablation): batches, loop trips and CAS traffic of k_finish's
claim/publish loop on the bench workload, cold table, one lane"""
import os, sys
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
for it in range(3):
    eng.clear(); eng.reset_stats()
    eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n)
    eng.sync(); st = eng.stats()
    print({k: st[k] for k in ("insert_path", "table_entries", "counted", "dbg0", "dbg1", "dbg2", "dbg3")},
          "trips/batch %.2f" % (st["dbg1"] / max(st["dbg0"], 1)), "reads/batch %.1f" % (st["insert_path"] / max(st["dbg0"], 1)), flush=True)
