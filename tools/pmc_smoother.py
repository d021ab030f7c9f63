"""Launches the finest-level fused RB-GS sweep a few times (ping-pong) so that rocprofv3 --pmc can attribute
HBM traffic to it:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_smoother.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
from evostencils_amd import abi, native, problems

level = int(sys.argv[1]) if len(sys.argv) > 1 else 12
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
kind = sys.argv[3] if len(sys.argv) > 3 else "poisson2d"          # poisson2d | poisson3d
smoother = sys.argv[4] if len(sys.argv) > 4 else "rbgs"            # rbgs | jacobi | residual
pd = (problems.poisson_3d if kind == "poisson3d" else problems.poisson_2d)(level - 1, level)
h = native.create(pd)
if smoother.startswith("block"):                                   # block211 etc.
    shape = tuple(int(c) for c in smoother[5:]) if len(smoother) > 5 else (2, 1, 1)
    op = abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, shape, 0.8)
elif smoother == "residual":
    op = abi.make_op(abi.OP_RESIDUAL, level, abi.F_RES, abi.F_SOL, abi.F_RHS)
else:
    op = abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS,
                     abi.PART_RED_BLACK if smoother == "rbgs" else abi.PART_SINGLE, (1, 1, 1), 1.0)
h.run_ops([op] * reps)
print("kernel", h.kernel_name(op), "launches", reps, "nodes", ((1 << level) + 1) ** pd.dimension)
