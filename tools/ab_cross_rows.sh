#!/bin/bash
# Developer tool: chunk heights of the cross-iteration kernel (EVOMG_XFUSE_ROWS) inside C2 solves, one bench run each.
# usage (GPU box): bash tools/ab_cross_rows.sh > gpurun_out/r04/cross_rows.txt
for rows in 0 48 64 80 96 112 128; do
  EVOMG_XFUSE_ROWS=$rows python bench.py --steps 12 --warmup 3 --population 0 --no-cpu-baseline --no-reference-size --no-configs --throughput-workers 0 2>/dev/null |
    python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('rows $rows: %.2f solves/s, solve %.3f ms, cross launch %.1f us, frac %.3f' % (d['value'], d['config']['solve_time_ms'], 1e3*r['ms_per_launch'], r['frac']))"
done
