"""Developer tool: per-op timings (HIP events, back-to-back launches) on one level of a problem.
usage: python tools/bench_ops.py poisson3d 8   |  poisson2d 11  |  helmholtz2d 11"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
from evostencils_amd import abi, native, problems

kind = sys.argv[1] if len(sys.argv) > 1 else "poisson3d"
level = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
pd = {"poisson2d": problems.poisson_2d, "poisson3d": problems.poisson_3d,
      "helmholtz2d": lambda a, b: problems.helmholtz_2d(a, b, k=0.625 * (1 << b))}[kind](level - 1, level)
h = native.create(pd)
one = (1, 1, 1)
d = pd.dimension
ops = [("jacobi", abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 0.8)),
       ("rbgs", abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, 1.0)),
       ("block(2,1,1)", abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 1, 1), 0.8)),
       ("block(2,2,1)", abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 2, 1), 0.8)),
       ("block(1,2,1)", abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1, 2, 1), 0.8)),
       ("residual", abi.make_op(abi.OP_RESIDUAL, level, abi.F_RES, abi.F_SOL, abi.F_RHS)),
       ("restrict", abi.make_op(abi.OP_RESTRICT, level - 1, abi.F_RHS, abi.F_RES)),
       ("prolong_add", abi.make_op(abi.OP_PROLONG_ADD, level, abi.F_SOL, abi.F_CORR, omega=1.0)),
       ("zero", abi.make_op(abi.OP_SET_ZERO, level, abi.F_RES))]
if d == 3:
    ops.insert(4, ("block(2,2,2)", abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 2, 2), 0.8)))
n = (1 << level) + 1
print(f"{kind} level {level}: {n}^{d} nodes, {n**d * (16 if h.is_complex else 8) / 1e6:.1f} MB per field")
for name, op in ops:
    b = h.bench_op(op, 3, reps)
    print(f"  {name:14s} {b.ms*1e3:9.1f} us  {b.gbps:7.0f} GB/s  ({b.gbps/80:5.1f} % of 8 TB/s)  [{h.kernel_name(op)}]", flush=True)
