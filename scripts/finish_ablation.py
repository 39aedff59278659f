"""finish-kernel ablation: cleared table (all keys new) vs warm table (lookups only), translate vs not"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import vfind_amd
from benchlib import synth
n = 10_000_000
ads = synth.adapters(0x5EED0002, 20)
gen = synth.Generator(0x5EED0002, ads)
rows_h, lens_h = gen.packed(0, n, 16)
rows = torch.from_numpy(rows_h).cuda(); lens = torch.from_numpy(lens_h.view(np.int16)).cuda()
for skip in (False, True):
    eng = vfind_amd.Engine(ads, skip_translation=skip, table_slots=1 << 22)
    for mode in ("cold", "warm", "warm"):
        if mode == "cold":
            eng.clear()
        eng.reset_stats(); eng.set_profiling(True)
        for _ in range(3):
            if mode == "cold":
                eng.clear()
            eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n)
        eng.sync(); st = eng.stats(); eng.set_profiling(False)
        print(f"skip_translation={skip} {mode}: finish {st['ms_finish']/3:.3f} ms scan {st['ms_scan']/3:.3f} filter {st['ms_filter']/3:.3f} dpw {st['ms_dp_window']/3:.3f} dp_rest {st['ms_dp_rest']/3:.3f} entries {st['table_entries']}", flush=True)
    eng.close()
