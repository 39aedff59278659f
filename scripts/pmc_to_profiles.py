#!/usr/bin/env python3
"""profiles/r02_pmc_sq.json from a `scripts/pmc_sq.sh` pass (rocprofv3 --kernel-trace --pmc SQ_* of bench.py, one batch lane).
usage: pmc_to_profiles.py counter_collection.csv kernel_trace.csv out.json [cfg4|cfg5: the shape scripts/cfg_shape.py ran]"""
import collections
import csv
import json
import re
import sys

MIN_GRID = 60000           # (k_log_apply: 65 536 threads at 2^21 slots)
SIMDS = 1024
CLOCK_HZ = 2.4e9            # MI355X_MICROARCH.md: max clock; the chip holds less under load, so the utilisation is a lower bound
ISSUE_CYCLES = 2.3          # one wave64 VALU instruction per ~2.3 cycles per SIMD at full rate


def short(name):
    return re.sub(r"\(.*", "", name).replace("void vf::", "").replace("vf::", "")


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if int(r.get("Grid_Size", 0)) < MIN_GRID:
        continue
    acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    if g >= MIN_GRID:
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"method": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                 "SQ_ACTIVE_INST_VALU -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extras with VFIND_RES_LANES=1 "
                 "(one batch lane: kernels of different batches do not overlap in the counters); per-launch means over the full-size launches",
       "definition": f"valu_util = SQ_INSTS_VALU x {ISSUE_CYCLES} cycles / ({SIMDS} SIMDs x kernel duration x {CLOCK_HZ / 1e9:.1f} GHz): the share of the chip's "
                     "full-rate VALU issue slots the kernel fills (a lower bound: the clock under load is below the 2.4 GHz used here; half-rate "
                     "instructions count once); wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave time parked on s_waitcnt / barriers)",
       "kernels": {}}
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = sum(dur[k]) / len(dur[k]) if dur.get(k) else None
    e = {"launches": max(len(v) for v in cs.values()), "avg_us": None if d is None else round(d, 1)}
    e.update({c: int(v) for c, v in m.items()})
    if d:
        e["valu_util"] = round(m.get("SQ_INSTS_VALU", 0) * ISSUE_CYCLES / (SIMDS * d * 1e-6 * CLOCK_HZ), 4)
    if m.get("SQ_WAVE_CYCLES"):
        e["wait_frac"] = round(m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 4)
        e["active_frac"] = round(m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 4)
    out["kernels"][k] = e
if len(sys.argv) > 4 and sys.argv[4]:
    out["method"] = out["method"].replace("python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extras",
                                          f"python3 scripts/cfg_shape.py {sys.argv[4]} (8 M resident reads a step, 1 warm-up + 10 steps)")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: (v.get("avg_us"), v.get("valu_util"), v.get("wait_frac")) for k, v in out["kernels"].items()}, indent=1))
