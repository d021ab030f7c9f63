# Developer tool: the solver loop's norm written straight into pinned host memory by the final reduction (EVOMG_NO_HOST_NORM=1: a
# device scalar + a copy launch per iteration, as before). usage (GPU box): bash tools/ab_host_norm.sh
for off in 0 1 0 1; do
  if [ $off = 1 ]; then export EVOMG_NO_HOST_NORM=1; else unset EVOMG_NO_HOST_NORM; fi
  echo "NO_HOST_NORM=$off"
  python bench.py --steps 10 --warmup 3 --population 64 --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
l = json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1])
r = l['reference_size']; p = l['population']
print('   headline %.2f (solve %.3f ms)' % (l['value'], l['config']['solve_time_ms']), 'throughput %.1f' % l['throughput_mode']['value'], '| reference size: %.1f solves/s (solve %.3f ms), %.1f in flight' % (r['value'], r['solve_time_ms'], r['in_flight']['value']), '| population in flight %.2f' % p['static_in_flight']['solves_per_s'])"
done
