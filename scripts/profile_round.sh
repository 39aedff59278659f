# the round's profile set: scripts/profile_round.sh r02  ->  gpurun_out/<tag>_* (copy the summaries into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r03}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/${tag}_trace.log 2>&1
f=$(find $R/gpurun_out/${tag}_trace -name "*kernel_trace.csv" | head -1)
k=$(find $R/gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1)
cp $k $R/gpurun_out/${tag}_kernel_stats.csv
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras (two batch lanes: kernels of consecutive batches overlap)"; echo; grep "^{" $R/gpurun_out/${tag}_trace.log | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('bench line of the profiled run: value', j['value'], 'M reads/s, ms_per_step', j['ms_per_step'], ', roofline', json.dumps(j['roofline']))
"; echo; python3 $R/scripts/summarize_prof.py $f; echo; echo '## mid-run timeline (us)'; echo '```'; python3 $R/scripts/timeline.py $f | head -40; echo '```'; } > $R/gpurun_out/${tag}_bench_kernel_trace.md
bash $R/scripts/pmc_sq.sh ${tag}_pmc_sq > $R/gpurun_out/${tag}_pmc_sq.txt 2>&1
bash $R/scripts/pmc_hbm.sh ${tag}_pmc_hbm > $R/gpurun_out/${tag}_pmc_hbm.txt 2>&1
tail -5 $R/gpurun_out/${tag}_bench_kernel_trace.md
