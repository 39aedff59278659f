# requests per load shape: bash scripts/ubench/row_stream.sh   (GPU box, repo root) -> gpurun_out/row_stream.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/row_stream.txt
$R/scripts/ubench/row_stream 10000000 5 > $OUT 2>&1
for c in TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/rs_$c -- $R/scripts/ubench/row_stream 10000000 1 > /dev/null 2>&1
  f=$(find /tmp/rs_$c -name "*counter_collection.csv" | head -1)
  python3 - $f $c >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]: acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k.replace("void ", ""): round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
  rm -rf /tmp/rs_$c
done
cat $OUT
