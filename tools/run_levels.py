"""Developer tool: V(1,1) RB-GS solve of 2D Poisson on the level range given. usage: python tools/run_levels.py lo hi"""
import sys
sys.path[:0] = ['/root/repo', '/root/repo/tests']
from evostencils_amd import abi, native, problems
from test_full_size_gpu import vcycle
lo, hi = int(sys.argv[1]), int(sys.argv[2])
pd = problems.poisson_2d(lo, hi)
ops = vcycle(lo, hi, 1, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.0))
h = native.create(pd)
h.evaluate(ops, 1, 0, 16)
out = h.evaluate(ops, 3, 0, 16)
print("levels %d..%d iterations %.0f rho %.4f time %.3f ms" % (lo, hi, out.n_iterations, out.conv_factor, out.time_ms))
