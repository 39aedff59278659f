# kernel-trace summaries of the bench workload under environment variants: scripts/prof_ab.sh tag "ENV=.. ENV=.." ["ENV=.."...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_$i -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-extras > $R/gpurun_out/${tag}_$i.log 2>&1
  f=$(find $R/gpurun_out/${tag}_$i -name "*kernel_trace.csv" | head -1)
  echo "== $envs"; python3 $R/scripts/summarize_prof.py $f | head -12
done
