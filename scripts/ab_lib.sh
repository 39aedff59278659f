# same-box A/B of library builds: scripts/ab_lib.sh ROUNDS "ENV=.. lib.so" "ENV=.. lib2.so" ...   (last word of each spec = library)
rounds=$1; shift
for i in $(seq $rounds); do
  for spec in "$@"; do
    lib=${spec##* }; envs=${spec% *}; [ "$envs" = "$spec" ] && envs=""
    env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras --lib $lib 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$spec', j['ms_per_step'], j['roofline_path']['kernel_ms_per_step'])
"
  done
done
