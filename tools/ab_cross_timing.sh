#!/bin/bash
# Developer tool: what bounds the cross-iteration kernel — builds with EVOMG_CROSS_TIMING (csrc/evomg_strip_cross.h; WRONG results,
# timing only) under build_exp/: t1 halo rows read from the chunk's own rows, t2 no loads of x / f in the row loop, t4 no stores, t8 no loads of the
# coarse correction, sums of those. ROWS="0 160": chunk heights (EVOMG_XFUSE_ROWS; 0 = the default, 80).
# usage (GPU box): bash tools/ab_cross_timing.sh > gpurun_out/r05_cross_timing.txt
for lib in "" build_exp/libevomg_t1.so build_exp/libevomg_t2.so build_exp/libevomg_t4.so build_exp/libevomg_t6.so build_exp/libevomg_t8.so build_exp/libevomg_t14.so; do
  [ -z "$lib" ] || [ -f "$lib" ] || continue
  for rows in ${ROWS:-0}; do
  EVOMG_XFUSE_ROWS=$rows EVOMG_LIB=${lib:+$PWD/$lib} python bench.py --steps 12 --warmup 3 --population 0 --no-cpu-baseline --no-reference-size --no-configs --throughput-workers 0 2>/dev/null |
    python -c "
import json,sys
t=sys.stdin.read()
try:
  d=json.loads([l for l in t.splitlines() if l.startswith('{')][-1]); r=d['roofline']
  print('lib ${lib:-default} rows $rows: cross launch %.1f us' % (1e3*r['ms_per_launch']))
except Exception as e:
  print('lib ${lib:-default} rows $rows: failed', e, t[-300:])"
  done
done
