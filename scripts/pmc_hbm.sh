# FETCH_SIZE and WRITE_SIZE passes (separate: they do not fit one pass) of the bench workload, one batch lane
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r2_pmc_hbm}
for c in FETCH_SIZE WRITE_SIZE; do
  VFIND_RES_LANES=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d ${OUT}_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extras > ${OUT}_$c.log 2>&1
done
f=$(find ${OUT}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find ${OUT}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 $R/scripts/pmc_hbm.py $f $w $OUT.json
