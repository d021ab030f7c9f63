"""Time of the coarse CG at 129^2 (the C2 coarsest grid) from a zero start on a random right-hand side.
usage: [EVOMG_LIB=...] python tools/cg129_time.py [lo]"""
import sys, time
sys.path[:0] = ['/root/repo', '/root/repo/tests']
import numpy as np
from evostencils_amd import abi, native, problems

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = (1 << lo) + 1
rng = np.random.default_rng(lo)
rhs = rng.standard_normal((n, n))
pd = problems.poisson_2d(lo, lo + 1)
h = native.create(pd)
h.set_field(lo, abi.F_CGS_RHS, rhs)
reps = 20
ops = [abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR), abi.make_op(abi.OP_COARSE_SOLVE, lo)] * reps
h.run_ops(ops)
h.get_field(lo, abi.F_CORR)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    h.run_ops(ops)
    h.get_field(lo, abi.F_CORR)
    best = min(best, (time.perf_counter() - t0) / reps)
its = h.last_coarse_iterations() if hasattr(h, "last_coarse_iterations") else None
print("%d^2: %.1f us per solve%s" % (n, best * 1e6, "" if not its else ", %d iterations -> %.2f us each" % (its, best * 1e6 / its)))
