for i in 1 2; do
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --population 0 --throughput-workers 12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['reference_size']; print('graph on ', r['value'], r['in_flight']['value'], r['solve_time_ms'])"
EVOMG_NO_GRAPH=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --population 0 --throughput-workers 12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['reference_size']; print('graph off', r['value'], r['in_flight']['value'], r['solve_time_ms'])"
done
