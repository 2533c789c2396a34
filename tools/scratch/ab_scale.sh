for r in 1 2 3; do for c in cfg4 cfg3 cfg2; do python3 tools/solve_loop.py $c 40; done; done
echo -n "K1 events "; SFMBA_K1_EVENTS=1 python3 tools/solve_loop.py cfg4 40
python3 tools/solve_loop.py cfg5 6 32
SFMBA_DEBUG=trace_timing=1 python3 tools/solve_loop.py cfg4 3 2>&1 | grep "sfmba: solve\|upload_x" | tail -4
for r in 1 2 3; do python3 bench.py --no-cpu-baseline --no-per-call 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('bench', round(d['value'],1), d['config']['ms_per_solve'], 'K1', round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac'],3))
"; done
