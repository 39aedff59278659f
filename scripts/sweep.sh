# bench.py under environment variants, one line each: scripts/sweep.sh "ENV=.. ENV=.." ...
for envs in "$@"; do
  env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$envs', j['ms_per_step'], j['roofline_path']['kernel_ms_per_step'])
"
done
