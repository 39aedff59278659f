"""per-kernel average of ONE rocprofv3 counter over the full-size launches of a bench run
usage: pmc_requests.py counter_collection.csv COUNTER  -> one JSON line {counter, kernels: {name: {avg, launches}}}"""
import collections
import csv
import json
import re
import sys

path, counter = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != counter:
        continue
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    acc[name].append((int(r.get("Grid_Size", 0)), float(r["Counter_Value"])))
out = {}
for k, v in acc.items():
    gmax = max(g for g, _ in v)
    full = [x for g, x in v if g * 2 >= gmax]          # the warm-up / tail launches over small inputs are left out
    if not k.startswith("vf::"):
        continue
    out[k] = {"avg": round(sum(full) / len(full), 1), "launches": len(full)}
print(json.dumps({"counter": counter, "kernels": out}))
