# Developer tool: restoring the initial fields in one launch (EVOMG_NO_RESET_BATCH=1: one memset / copy per field) on the
# reference-size section of the bench line and on the host overhead of one evaluation. usage (GPU box): bash tools/ab_reset_batch.sh
for off in 0 1 0 1; do
  if [ $off = 1 ]; then export EVOMG_NO_RESET_BATCH=1; else unset EVOMG_NO_RESET_BATCH; fi
  echo "NO_RESET_BATCH=$off"; python tools/host_overhead.py 5 9 2>&1 | tail -3 | tr '\n' ';'; echo
  python bench.py --steps 10 --warmup 3 --population 0 --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
l = json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1])
r = l['reference_size']
print('   headline %.2f' % l['value'], 'throughput %.1f' % l['throughput_mode']['value'], '| reference size: %.1f solves/s, %.1f with twelve in flight' % (r['value'], r['in_flight']['value']))"
done
