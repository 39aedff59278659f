import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
import vfind_amd
from vfind_amd import synth
n = 10_000_000
ads = synth.adapters(0x5EED0002, 20)
gen = synth.Generator(0x5EED0002, ads)
rows_h, lens_h = gen.packed(0, n, 16)
rows = torch.from_numpy(rows_h).cuda(); lens = torch.from_numpy(lens_h.view(np.int16)).cuda()
for flags in ("0", "0x200", "0x100"):
    os.environ["VFIND_DEBUG_FLAGS"] = flags
    eng = vfind_amd.Engine(ads, table_slots=1 << 22)
    eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n); eng.sync()
    eng.reset_stats(); eng.set_profiling(True)
    for _ in range(3):
        eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n)
    eng.sync(); st = eng.stats()
    print(f"flags={flags} warm: finish {st['ms_finish']/3:.3f} ms", flush=True)
    eng.close()
