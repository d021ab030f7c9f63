"""Developer tool: the 33^3 coarse solve (evomg_cg_multi_wg3d) inside a V(2,2) solve of 3D Poisson, levels 5..7 (129^3):
time per call, Krylov iterations per call, microseconds per iteration. usage: python tools/bench_coarse3d.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
from evostencils_amd import abi, native, problems
from test_full_size_gpu import vcycle

lo, hi = 5, int(sys.argv[1]) if len(sys.argv) > 1 else 7
pd = problems.poisson_3d(lo, hi)
h = native.create(pd, 0, jacobi_semantics="textbook")
ops = vcycle(lo, hi, 2, 2, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 1, 1), 0.8))
h.evaluate(ops, 1, 0, 0)
best = None
for _ in range(5):
    h.profile_begin(abi.OP_COARSE_SOLVE, lo)
    res = h.evaluate(ops, 1, 0, 0)
    b = h.profile_end()
    k = h.coarse_iterations()
    cur = (b.ms * b.launches * 1e3 / k, b.ms * 1e3, k / b.launches, res.time_ms)
    best = cur if best is None or cur[0] < best[0] else best
print("coarse solve: %.3f us per CG iteration, %.1f us per call, %.1f iterations per call; whole solve %.3f ms" % best)
