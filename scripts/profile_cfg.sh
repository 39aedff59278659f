# rocprofv3 kernel stats + filter class histogram of one of the other BASELINE shapes (scripts/cfg_shape.py):
#   bash scripts/profile_cfg.sh r04 cfg4      ->  gpurun_out/r04_cfg4_kernel_stats.csv, gpurun_out/r04_cfg4_kernel_trace.md
# (copy what is to be judged into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shape=$2
D=$R/gpurun_out/${tag}_${shape}_trace
CFG_PROF=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/scripts/cfg_shape.py $shape > $R/gpurun_out/${tag}_${shape}_trace.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_${shape}_trace.log; exit 1; }
f=$(find $D -name "*kernel_trace.csv" | head -1)
k=$(find $D -name "*kernel_stats.csv" | head -1)
cp $k $R/gpurun_out/${tag}_${shape}_kernel_stats.csv
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 scripts/cfg_shape.py $shape   (8 M resident reads a step, 3 warm-up + 10 timed steps, two batch lanes: kernels of consecutive batches overlap)"
  echo
  echo "run line: $(grep "M reads/s" $R/gpurun_out/${tag}_${shape}_trace.log | tail -1)"
  echo
  python3 $R/scripts/summarize_prof.py $f
  echo
  echo "## filter classes of one batch of 2 M reads of the same input (VFIND_DEBUG_CLASSES=1, a separate un-profiled run)"
  echo '```'
  cd $R && VFIND_DEBUG_CLASSES=1 CFG_WARM=0 timeout -k 10 300 python3 scripts/cfg_shape.py $shape - 2000000 2>&1 | grep "vfind classes" | head -1
  echo '```'
} > $R/gpurun_out/${tag}_${shape}_kernel_trace.md
rm -rf $D
tail -30 $R/gpurun_out/${tag}_${shape}_kernel_trace.md
