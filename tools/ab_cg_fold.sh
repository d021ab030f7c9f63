for off in 0 1 0 1; do
  if [ $off = 1 ]; then export EVOMG_NO_CG_FOLD=1; else unset EVOMG_NO_CG_FOLD; fi
  python bench.py --steps 10 --warmup 3 --population 64 --no-configs --no-reference-size --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
l = json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1])
print('NO_CG_FOLD=$off', 'headline %.2f' % l['value'], 'throughput_mode %.1f' % l['throughput_mode']['value'], 'population in flight %.2f' % l['population']['static_in_flight']['solves_per_s'], 'sequential %.2f' % l['population']['static']['solves_per_s'])"
done
