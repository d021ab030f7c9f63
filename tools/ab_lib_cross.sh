#!/bin/bash
# Developer tool: the cross-iteration launch inside C2 solves with several builds of the library, interleaved on one box.
# usage (GPU box): bash tools/ab_lib_cross.sh default build_exp/libevomg_a.so ...     (ENVS="A=1 B=2": settings tried per build)
for rep in 1 2; do
  for lib in "$@"; do
    for envs in "-" ${ENVS}; do
    ( if [ "$lib" != default ]; then export EVOMG_LIB=$PWD/$lib; fi
      if [ "$envs" != "-" ]; then export $envs; fi
      python bench.py --steps 12 --warmup 3 --population 0 --no-cpu-baseline --no-reference-size --no-configs --throughput-workers 0 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read()
try:
  d=json.loads([l for l in t.splitlines() if l.startswith('{')][-1]); r=d['roofline']
  print('rep $rep lib $lib env $envs: value %.1f solves/s, cross launch %.1f us frac %.3f' % (d['value'], 1e3*r['ms_per_launch'], r['frac']))
except Exception as e:
  print('lib $lib: failed', e, t[-300:])" )
    done
  done
done
