"""summarise a rocprofv3 --pmc counter_collection.csv: per kernel, mean counter value over its big launches"""
import csv, sys, collections, re
path = sys.argv[1]
min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(path) as f:
    for row in csv.DictReader(f):
        if int(row.get("Grid_Size", 0)) < min_grid:
            continue
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void vf::", "").replace("vf::", "")
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    n = max(len(v) for v in cs.values())
    print(f"{k}  (launches {n})")
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} {sum(v)/len(v):16.0f}")
