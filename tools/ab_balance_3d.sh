# Developer tool: the two round-4 launch-geometry changes of the 3D tiles on C4 (513^3), interleaved on one box:
# strips of equal width (EVOMG_NO_XBALANCE3D=1 switches back to 124/120-column strips; A/B only — the 3D loop norm's canonical
# order is defined on the equal strips) and z-chunk depth by workgroup balance (EVOMG_NO_BALANCE3D=1: 16 planes as before;
# EVOMG_BALANCE3D=<n>: deepest chunk allowed). Then back-to-back timings of the 513^3 and 257^3 kernels.
# usage (GPU box): bash tools/ab_balance_3d.sh
run() { RUN_CONFIG_REPEATS=3 python tools/run_config.py c4 2>&1 | grep "best of" ; }
for round in 1 2; do
  echo "old strips, depth 16";            EVOMG_NO_XBALANCE3D=1 EVOMG_NO_BALANCE3D=1 run
  echo "equal strips, depth 16";          EVOMG_NO_BALANCE3D=1 run
  echo "old strips, balanced depth";      EVOMG_NO_XBALANCE3D=1 run
  echo "equal strips, balanced depth";    run
done
for lvl in 9 8 7; do
for off in 1 0; do
if [ $off = 1 ]; then export EVOMG_NO_BALANCE3D=1 EVOMG_NO_XBALANCE3D=1; else unset EVOMG_NO_BALANCE3D EVOMG_NO_XBALANCE3D; fi
LVL=$lvl OFF=$off python - <<PY
import os, sys
sys.path[:0]=["/root/repo","/root/repo/tests"]
from evostencils_amd import abi, native, problems
L = int(os.environ["LVL"])
pd = problems.poisson_3d(L - 3, L)
h = native.create(pd)
for name, op in (("jacobi", abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1,1,1), 0.8)),
                 ("rbgs", abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1,1,1), 1.0)),
                 ("block211", abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2,1,1), 0.8)),
                 ("residual", abi.make_op(abi.OP_RESIDUAL, L, abi.F_RES, abi.F_SOL, abi.F_RHS))):
    b = sorted((h.bench_op(op, warmup=3, reps=20) for _ in range(3)), key=lambda b: b.ms)[1]
    print("level", L, "old geometry" if os.environ["OFF"] == "1" else "balanced", name, "%.1f us  frac %.3f" % (1e3*b.ms, b.gbps/8000))
PY
done
done
