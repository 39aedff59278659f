#!/usr/bin/env python3
"""text timeline of a rocprofv3 kernel trace: scripts/timeline.py kernel_trace.csv [first_us] [span_us]
one line per launch of the full-size kernels: start, duration, stream (queue), name; plus the mean period of k_finish"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void vf::", "").replace("vf::", "")
    if "rocclr" in name:
        continue
    ev.append(((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
ev.sort()
fin = [e for e in ev if e[3].startswith("k_finish") and e[4] > 100000]
if len(fin) > 3:
    per = (fin[-1][0] - fin[2][0]) / (len(fin) - 3)
    print(f"k_finish period over the last {len(fin) - 3} launches: {per:.1f} us; mean duration {sum(e[1] for e in fin[2:]) / len(fin[2:]):.1f} us")
first = float(sys.argv[2]) if len(sys.argv) > 2 else fin[len(fin) // 2][0] - 200
span = float(sys.argv[3]) if len(sys.argv) > 3 else 2200
for s, d, q, name, g in ev:
    if first <= s <= first + span:
        print(f"{s - first:9.1f} +{d:7.1f}  q{q}  {name[:28]:28s} {'#' * int(d / 12)}")
