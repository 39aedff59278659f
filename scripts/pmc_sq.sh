# SQ counter pass of the bench workload, one batch lane (kernels of different batches do not overlap in the counters)
#   bash scripts/pmc_sq.sh r04_pmc_sq            the headline workload (bench.py)
#   bash scripts/pmc_sq.sh r04_cfg4_pmc_sq cfg4  one of the other BASELINE shapes (scripts/cfg_shape.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r2_pmc_sq}
if [ -n "$2" ]; then PROG="$R/scripts/cfg_shape.py $2"; export CFG_PROF=0 CFG_WARM=1; else PROG="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extras"; fi
VFIND_RES_LANES=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -- python3 $PROG > $OUT.log 2>&1
f=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 $R/scripts/pmc_summary.py $f
t=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/pmc_to_profiles.py $f $t $OUT.json "$2"
