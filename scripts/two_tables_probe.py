"""experiment: does a second table (here: a second context) used in turn remove the clear's sync point between steps?
usage: two_tables_probe.py [lanes_per_engine]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lanes = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ["VFIND_RES_LANES"] = lanes
import numpy as np, torch
import vfind_amd
from benchlib import synth
n, L, seed = 10_000_000, 150, 0x5EED0002
ads = synth.adapters(seed, 20)
gen = synth.Generator(seed, ads, read_len=L, n_variants=1_000_000, sub_rate=0.005, indel_rate=0.0005)
rows_h, lens_h = gen.packed(0, n, 16)
dev = torch.device("cuda", 0)
rows = torch.from_numpy(rows_h).to(dev); lens = torch.from_numpy(lens_h.view(np.int16)).to(dev)
for n_eng in (1, 2):
    engs = [vfind_amd.Engine(ads, device=0, table_slots=1 << 21, arena_bytes=1 << 30) for _ in range(n_eng)]
    def step(k):
        e = engs[k % n_eng]
        e.clear(); e.process_device(rows.data_ptr(), gen.row_dwords, lens.data_ptr(), n)
    for k in range(4): step(k)
    for e in engs: e.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for k in range(K): step(k)
    for e in engs: e.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"lanes/engine {lanes}, engines {n_eng}: {dt / K * 1e3:.4f} ms/step, {n * K / dt / 1e9:.3f} G reads/s, entries {[e.stats()['table_entries'] for e in engs]}")
    for e in engs: e.close()
