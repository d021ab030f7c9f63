# In-flight workers x hardware queues sweep on the bench line (run on the GPU box: bash tools/ab_workers.sh). Final build of round 3:
# 8/16 245, 12/16 267, 16/16 216, 16/24 221, 20/24 268 solves/s (isolated 68.6-69.0 throughout): 12 workers on 16 queues stays.
# Round 4 (fewer launches per iteration): see profiles/r04_workers_raw.txt
for cfg in "8 16" "12 16" "16 16" "16 24" "20 24" "24 24" "24 32"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --population 0 --no-configs --throughput-workers $1 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1])
print('workers $1 queues $2: C2', round(d['throughput_mode']['value'],1), 'isolated', round(d['value'],1), '| 513^2 in flight', round(d['reference_size']['in_flight']['value'],1))"
done
