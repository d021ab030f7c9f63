# HISTORICAL (round 3): the libraries under build/ were -DEVOMG_EXPERIMENT_NT_X builds of commit 3250de9's strip kernels; the
# placement they measured (profiles/r03_nt_placement.md) is what evomg_strip2d<.., NT_XY> now does. Kept as the record of the method.
# per-op timings under the four non-temporal placements (see tools/ab_nt.sh)
for lvl in 12 13; do
  for pol in "A:f:" "B:none:0" "C:xy:0:ntx" "D:all:1:ntx"; do
    IFS=: read tag name nt lib <<< "$pol"
    unset EVOMG_LIB EVOMG_STRIP_NT
    [ -n "$nt" ] && export EVOMG_STRIP_NT=$nt
    [ -n "$lib" ] && export EVOMG_LIB=$PWD/build/libevomg_$lib.so
    echo "== level $lvl policy $tag (NT on $name)"
    python tools/bench_ops.py poisson2d $lvl 30 2>&1 | grep -E "jacobi|rbgs|residual|prolong_add|block\(2,1"
  done
done
