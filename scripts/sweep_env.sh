# interleaved sweep of one environment variable over the headline bench (no CPU legs): bash scripts/sweep_env.sh VAR "v1 v2 ..." [rounds] [extra bench args]
# prints "VAR=value ms_per_step value" per run; run it on the GPU box from the repo root
VAR=$1; VALS=$2; ROUNDS=${3:-2}; shift 3 2>/dev/null
for r in $(seq 1 $ROUNDS); do
  for v in $VALS; do
    line=$(env $VAR=$v timeout -k 10 300 python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-parity --no-extras "$@" 2>/dev/null | grep '^{' | tail -1)
    rc=$?
    echo "$VAR=$v $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["ms_per_step"], j["value"], j["roofline_path"]["kernel_ms_per_step"])' 2>/dev/null)"
  done
done
