for keep in 0 1; do for b in 2 3 4 5 6; do
 VFIND_BENCH_KEEP_TABLE=$keep VFIND_RES_LANES=1 VFIND_FINISH_BLOCKS_PER_CU=$b python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('keep=$keep blocks=$b one-lane', j['ms_per_step'], j['roofline_path']['kernel_ms_per_step'])
"
done; done
