#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel, stats over the FULL-SIZE launches only
(the small parity-gate / test launches of the same kernels are listed separately)."""
import csv
import sys
from collections import defaultdict


def main(path, min_grid=100_000):
    rows = list(csv.DictReader(open(path)))
    big, small = defaultdict(list), defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0)
        (big if grid >= min_grid else small)[name].append((dur, grid, r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?")))
    print(f"| kernel | launches | avg us | min us | max us | grid (threads) | VGPR | LDS B |")
    print("|---|---|---|---|---|---|---|---|")
    for name, v in sorted(big.items(), key=lambda kv: -sum(d for d, *_ in kv[1])):
        d = [x[0] for x in v]
        print(f"| `{name}` | {len(d)} | {sum(d)/len(d):.1f} | {min(d):.1f} | {max(d):.1f} | {v[0][1]} | {v[0][2]} | {v[0][3]} |")
    print()
    print("small launches (parity gate, memset/copy helpers):")
    for name, v in sorted(small.items(), key=lambda kv: -sum(d for d, *_ in kv[1])):
        d = [x[0] for x in v]
        print(f"- `{name}`: {len(d)} launches, avg {sum(d)/len(d):.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 100_000)
