#!/bin/bash
# Developer tool: throughput mode (tools/run_throughput.py) with several builds of the library, interleaved on one box.
# usage (GPU box): bash tools/ab_lib_tput.sh "<workers> <solves>" default build_exp/libevomg_a.so build_exp/libevomg_b.so ...
ws=${1:-"16 256"}; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset EVOMG_LIB; else export EVOMG_LIB=$PWD/$lib; fi
    echo "rep $rep lib $lib: $(python tools/run_throughput.py $ws 2>/dev/null | grep throughput | cut -d';' -f1)"
  done
done
