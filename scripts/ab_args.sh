# same-box A/B of bench.py argument variants: scripts/ab_args.sh ROUNDS "args.." "args.." ...
rounds=$1; shift
for i in $(seq $rounds); do
  for spec in "$@"; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras $spec 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$spec', j['ms_per_step'], j['roofline_path']['kernel_ms_per_step'])
"
  done
done
