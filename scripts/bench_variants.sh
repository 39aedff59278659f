# bench.py under a few environment switches, one line each (value, ms/step, per-kernel event ms/step)
set -e
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline_path']['kernel_ms_per_step'])
"; }
for v in "$@"; do run $v; done
