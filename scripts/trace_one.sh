# scripts/trace_one.sh <tag> [env assignments...]: kernel trace of the bench (no extras) -> gpurun_out/<tag>_trace.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/${tag}_trace.log 2>&1
f=$(find $R/gpurun_out/${tag}_trace -name "*kernel_trace.csv" | head -1)
{ echo "# $* rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"; grep "^{" $R/gpurun_out/${tag}_trace.log | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('bench line of the profiled run: value', j['value'], 'M reads/s, ms_per_step', j['ms_per_step'])
"; python3 $R/scripts/summarize_prof.py $f; echo '## mid-run timeline (us)'; python3 $R/scripts/timeline.py $f | head -60; } > $R/gpurun_out/${tag}_trace.md
rm -rf $R/gpurun_out/${tag}_trace
