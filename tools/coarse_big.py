import sys, dataclasses
sys.path[:0]=['/root/repo','/root/repo/tests']
from evostencils_amd import abi, native, problems
from test_full_size_gpu import vcycle
for name, pd, lo, hi in (("3D 257^3, coarse 65^3", problems.poisson_3d(6, 8), 6, 8), ("2D 2049^2, coarse 257^2", problems.poisson_2d(8, 11), 8, 11),
                         ("2D 4097^2, coarse 513^2", problems.poisson_2d(9, 12), 9, 12)):
    pd = dataclasses.replace(pd, max_iterations=3)
    h = native.create(pd)
    ops = vcycle(lo, hi, 1, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.0))
    out = h.evaluate(ops, 1, 0, 16)
    print(name, "iterations", out.n_iterations, "time %.1f ms -> %.1f ms per V-cycle" % (out.time_ms, out.time_ms / max(out.n_iterations, 1)), flush=True)
    h.close()
