# the other BASELINE config shapes at 10 M reads on one GPU, under environment variants: scripts/sweep_cfg.sh "ENV=.." ...
for envs in "$@"; do
  for cfg in "cfg4 --sub-rate 0.15 --indel-rate 0 --seed 0x5EED0004" "cfg5 --read-len 300 --adapter-len 30 --variants 10000000 --seed 0x5EED0005"; do
    set -- $cfg; name=$1; shift
    env $envs python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$envs $name', j['ms_per_step'], j['value'], j['roofline_path']['kernel_ms_per_step'], j['parity_gate'])
"
  done
done
