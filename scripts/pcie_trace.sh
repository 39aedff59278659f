# copy + kernel timeline of the pinned batch API leg -> gpurun_out/<tag>_pcie_copies.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r3}
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/${tag}_pcie -- python3 $R/scripts/pcie_diag.py trace > $R/gpurun_out/${tag}_pcie_trace.log 2>&1
python3 - <<PY > $R/gpurun_out/${tag}_pcie_copies.txt
import csv, glob
f = glob.glob("$R/gpurun_out/${tag}_pcie/**/*memory_copy_trace.csv", recursive=True)
k = glob.glob("$R/gpurun_out/${tag}_pcie/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
print("columns:", list(rows[0].keys()) if rows else None)
big = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000]
big.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(big[0]["Start_Timestamp"]) if big else 0
prev_end = None
for r in big[-40:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f" gap {(s - prev_end) / 1e3:8.1f} us"
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  {r.get('Direction', r.get('Name', ''))}{gap}")
    prev_end = e
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in big)
span = int(big[-1]["End_Timestamp"]) - t0 if big else 1
print(f"copies > 100 us: {len(big)}, busy {busy / 1e6:.2f} ms of span {span / 1e6:.2f} ms ({100 * busy / span:.0f} %)")
PY
rm -rf $R/gpurun_out/${tag}_pcie
