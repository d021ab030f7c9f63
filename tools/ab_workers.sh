# In-flight workers x hardware queues sweep on the bench line (run on the GPU box: bash tools/ab_workers.sh). Final build of round 3:
# 8/16 245, 12/16 267, 16/16 216, 16/24 221, 20/24 268 solves/s (isolated 68.6-69.0 throughout): 12 workers on 16 queues stays.
for cfg in "8 16" "12 16" "16 16" "16 24" "20 24"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --population 0 --no-reference-size --throughput-workers $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('workers $1 queues $2:', round(d['throughput_mode']['value'],1), 'isolated', round(d['value'],1))"
done
