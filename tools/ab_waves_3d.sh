# Developer tool: four- or eight-wave tile workgroups on the 257^3 and 129^3 levels (EVOMG_TILE3D_WAVES), back to back.
for lvl in 8 7; do
for wv in 0 8 4; do
if [ $wv = 0 ]; then unset EVOMG_TILE3D_WAVES; else export EVOMG_TILE3D_WAVES=$wv; fi
LVL=$lvl WV=$wv python - <<PY
import os, sys
sys.path[:0]=["/root/repo","/root/repo/tests"]
from evostencils_amd import abi, native, problems
L = int(os.environ["LVL"])
pd = problems.poisson_3d(L - 2, L)
h = native.create(pd)
for name, op in (("jacobi", abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1,1,1), 0.8)),
                 ("rbgs", abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1,1,1), 1.0)),
                 ("block211", abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2,1,1), 0.8)),
                 ("residual", abi.make_op(abi.OP_RESIDUAL, L, abi.F_RES, abi.F_SOL, abi.F_RHS))):
    b = sorted((h.bench_op(op, warmup=3, reps=20) for _ in range(3)), key=lambda b: b.ms)[1]
    print("level", L, "waves", os.environ["WV"], name, "%.1f us  frac %.3f" % (1e3*b.ms, b.gbps/8000))
PY
done
done
