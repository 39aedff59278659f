# Memory-side REQUEST counters per kernel, ONE counter per rocprofv3 pass (the TCC/TCP groups asked for together
# abort with "Request exceeds the capabilities of the hardware to collect", rounds 1-2).
# usage: bash scripts/pmc_requests.sh <tag> <counter> [<counter> ...]      (on the GPU box, from the repo root)
# Every pass runs under its own timeout; a pass that times out ends the script (a hung GPU step is followed by none),
# a pass that fails otherwise (unknown counter name, profile refused) is recorded and the next one runs.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
OUT=$R/gpurun_out/${tag}
mkdir -p $OUT
LANES=${LANES:-1}
STEPS=${STEPS:-3}
for c in "$@"; do
  echo "== $c (lanes $LANES, steps $STEPS) $(date +%T)" | tee -a $OUT/passes.txt
  VFIND_RES_LANES=$LANES timeout -k 10 ${PASS_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-parity --no-extras > $OUT/$c.log 2>&1
  rc=$?
  echo "   rc=$rc $(date +%T)" | tee -a $OUT/passes.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "   TIMEOUT on $c: stopping" | tee -a $OUT/passes.txt; tail -5 $OUT/$c.log; exit 0; fi
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/scripts/pmc_requests.py $f $c >> $OUT/summary.txt 2>&1; else tail -3 $OUT/$c.log | tee -a $OUT/passes.txt; fi
  rm -rf $OUT/$c
done
cat $OUT/summary.txt
