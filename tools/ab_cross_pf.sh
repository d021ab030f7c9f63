#!/bin/bash
# Developer tool: prefetch depth (compile-time EVOMG_CROSS_PF: builds under build_exp/, see profiles/r05_cross_kernel.md) x
# chunk height (EVOMG_XFUSE_ROWS) of the cross-iteration kernel inside C2 solves, one short bench run each.
# usage (GPU box): bash tools/ab_cross_pf.sh > gpurun_out/r05_cross_pf.txt
for lib in "" build_exp/libevomg_xpf2.so build_exp/libevomg_xpf1.so; do
  for rows in 0 48 64 80; do
    EVOMG_LIB=${lib:+$PWD/$lib} EVOMG_XFUSE_ROWS=$rows python bench.py --steps 12 --warmup 3 --population 0 --no-cpu-baseline --no-reference-size --no-configs --throughput-workers 0 2>/dev/null |
      python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('lib ${lib:-default} rows $rows: %.2f solves/s, solve %.3f ms, cross launch %.1f us, frac %.3f' % (d['value'], d['config']['solve_time_ms'], 1e3*r['ms_per_launch'], r['frac']))"
  done
done
