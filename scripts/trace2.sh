cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r2_trace}
shift
env "$@" rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-extras > $OUT.log 2>&1
tail -c 400 $OUT.log
