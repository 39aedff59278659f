"""vf_finish on a big table: ~10^7 distinct keys (BASELINE configs[4] shape: 300 bp reads, 30 bp adapters, flat 10^7-variant
library so that nearly every read brings a new key), device result build timed by the library (ms_result_build)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vfind_amd
from benchlib import synth
n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
chunk = 4_000_000
ads = synth.adapters(0x5EED0005, 30)
gen = synth.Generator(0x5EED0005, ads, read_len=300, n_variants=10_000_000, zipf_s=0.0)
eng = vfind_amd.Engine(ads, table_slots=1 << 25, arena_bytes=4 << 30)
for lo in range(0, n_total, chunk):
    rows_h, lens_h = gen.packed(lo, chunk, 16)
    rows = torch.from_numpy(rows_h).cuda(); lens = torch.from_numpy(lens_h.view(np.int16)).cuda()
    eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), chunk)
    eng.sync()
    del rows, lens, rows_h, lens_h
for it in range(2):
    t0 = time.perf_counter()
    keys, offs, cnts = eng.finish_arrays()
    dt = time.perf_counter() - t0
    st = eng.stats()
    print(f"rows {len(cnts)}  key bytes {int(offs[-1])}  vf_finish {st['ms_result_build']:.1f} ms (wall incl. numpy copies {dt*1e3:.0f} ms)  "
          f"sorted: {bool((cnts[:-1] >= cnts[1:]).all())}  sum counts {int(cnts.sum())} = counted {st['counted']}", flush=True)
eng.close()
