# Developer tool: the 33^3 coarse solve (evomg_cg_multi_wg3d), its eight workgroups on one XCD (default) or one per XCD.
# usage (GPU box): bash tools/ab_cg_poll.sh
for v in a c a c; do
  case $v in a) unset EVOMG_CG_SPREAD_XCDS; echo "one XCD";; c) export EVOMG_CG_SPREAD_XCDS=1; echo "one workgroup per XCD";; esac
  python tools/bench_coarse3d.py
done
