"""the same batch counted again and again into a cleared table: every pass must give the same (key, count) set and the
counts must sum to the reads the kernel says it counted (a race in the table kernel shows up as a pass that differs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("VFIND_STRESS_LIB"):
    from vfind_amd import _lib
    _lib.SO_PATH = os.path.abspath(os.environ["VFIND_STRESS_LIB"])
import numpy as np, torch
import vfind_amd
from benchlib import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
variants = int(sys.argv[2]) if len(sys.argv) > 2 else 30_000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ads = synth.adapters(0x5EED0002, 20)
gen = synth.Generator(0x5EED0002, ads, n_variants=variants)
rows_h, lens_h = gen.packed(0, n, 16)
rows = torch.from_numpy(rows_h).cuda(); lens = torch.from_numpy(lens_h.view(np.int16)).cuda()
eng = vfind_amd.Engine(ads, table_slots=1 << 21)
first = None
bad = 0
for it in range(passes):
    eng.clear()
    c0 = eng.stats()["counted"]
    eng.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n)
    eng.sync()
    counted = eng.stats()["counted"] - c0
    keys, offs, cnts = eng.finish_arrays()
    tot = int(np.asarray(cnts).sum())
    sig = (len(cnts), tot, int((np.asarray(cnts).astype(np.uint64) * (np.arange(len(cnts), dtype=np.uint64) + 1)).sum()))
    if first is None:
        first = sig
    if tot != counted or sig != first:
        bad += 1
        print(f"pass {it}: rows {sig[0]} sum {tot} counted {counted} signature {sig[2]} (first pass: {first})")
print(f"{passes} passes of {n} reads: {bad} differ")
sys.exit(1 if bad else 0)
