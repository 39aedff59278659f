# copy + kernel timeline of the FASTQ path (vf_run_fastq): do the host-to-device copies of chunk k+1 run under the kernels of chunk k?
#   bash scripts/fastq_trace.sh r04 8000000 16     ->  gpurun_out/r04_fastq_timeline.md   (copy into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r04}; reads=${2:-8000000}; nt=${3:-16}
D=$R/gpurun_out/${tag}_fastq_trace
python3 $R/scripts/fastq_diag.py $reads $nt /tmp/vfind_trace.fq.gz > /dev/null 2>&1        # writes the file, warms the page cache
timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -- python3 $R/scripts/fastq_diag.py $reads $nt /tmp/vfind_trace.fq.gz > $R/gpurun_out/${tag}_fastq_trace.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_fastq_trace.log; exit 1; }
python3 - <<PY > $R/gpurun_out/${tag}_fastq_timeline.md
import csv, glob
c = glob.glob("$D/**/*memory_copy_trace.csv", recursive=True)
k = glob.glob("$D/**/*kernel_trace.csv", recursive=True)
copies = [r for r in csv.DictReader(open(c[0]))] if c else []
kern = [r for r in csv.DictReader(open(k[0]))] if k else []
h2d = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in copies if "HOST_TO_DEVICE" in r.get("Direction", r.get("Name", "")) and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 20000))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vf::", ""), r.get("Queue_Id", "?")) for r in kern if "vf::" in r["Kernel_Name"]))
# the second run_fastq of the process is the timed one: take the last 60 % of the trace
t_lo = h2d[0][0] + (h2d[-1][1] - h2d[0][0]) * 2 // 5 if h2d else 0
h2d = [x for x in h2d if x[0] >= t_lo]
ks = [x for x in ks if x[0] >= t_lo]
def union(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
def overlap(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo: tot += hi - lo
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
uc, uk = union(h2d), union([(s, e) for s, e, *_ in ks])
span = (max(uc[-1][1], uk[-1][1]) - min(uc[0][0], uk[0][0])) if uc and uk else 1
tc, tk = sum(e - s for s, e in uc), sum(e - s for s, e in uk)
both = overlap(uc, uk)
print("# FASTQ path: host-to-device copies against kernels (round 4)")
print()
print("\`rocprofv3 --kernel-trace --memory-copy-trace -- python3 scripts/fastq_diag.py $reads $nt\` (scripts/fastq_trace.sh): vf_run_fastq on a multi-member gzip FASTQ of the bench's read model,")
print("queue_len 4, n_threads $nt; the second (timed) pass of the process.  Run line: " + next((l for l in open("$R/gpurun_out/${tag}_fastq_trace.log").read().splitlines() if l.startswith("fastq_diag:")), "?"))
print()
print(f"| span | copies busy (union) | kernels busy (union) | copy time under a kernel | kernel time under a copy |")
print(f"|---|---|---|---|---|")
print(f"| {span / 1e6:.1f} ms | {tc / 1e6:.1f} ms ({100 * tc / span:.0f} %) | {tk / 1e6:.1f} ms ({100 * tk / span:.0f} %) | {100 * both / max(tc, 1):.0f} % | {100 * both / max(tk, 1):.0f} % |")
print()
print("Mid-run window (us from its start): every text copy (H2D) and every kernel, with its hardware queue - copies of the next chunk start while the kernels of the chunk before are still running on another slot's stream:")
print("\`\`\`")
ev = [(s, e, "H2D text copy", "dma") for s, e in h2d] + list(ks)
ev.sort()
mid = ev[len(ev) // 2][0] if ev else 0
for s, e, name, q in ev:
    if mid <= s <= mid + 6000000:
        print(f"{(s - mid) / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  q{q:>4}  {name[:30]:30s} {'#' * min(int((e - s) / 25000), 60)}")
print("\`\`\`")
PY
rm -rf $D
head -12 $R/gpurun_out/${tag}_fastq_timeline.md
