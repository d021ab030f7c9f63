"""Developer tool: the one-wave coarse CG against the oracle on random right-hand sides, fused (COPY + SET_ZERO + solve in
one launch) and unfused, from zero and from a random initial guess."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import numpy as np
from evostencils_amd import abi, native, problems
from oracle import oracle

for lo in (1, 2, 3, 4, 5):
    pd = problems.poisson_2d(lo, lo + 1)
    n = (1 << lo) + 1
    rng = np.random.default_rng(lo)
    for mode in ("fused", "zero-unfused", "nonzero"):
        hg, ho = native.create(pd), oracle.create(pd)
        rhs = rng.standard_normal((n, n))
        x0 = rng.standard_normal((n, n))
        x0[0, :] = x0[-1, :] = x0[:, 0] = x0[:, -1] = 0
        for h in (hg, ho):
            h.set_field(lo, abi.F_RHS, rhs)
            h.set_field(lo, abi.F_SOL, x0)
        copy = abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS)
        zero = abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR)
        solve = abi.make_op(abi.OP_COARSE_SOLVE, lo)
        if mode == "fused":
            progs = [[copy, zero, solve]]
        elif mode == "zero-unfused":
            progs = [[copy], [zero], [solve]]
        else:
            progs = [[copy], [solve]]
        for prog in progs:
            hg.run_ops(prog)
            ho.run_ops(prog)
        for f in (abi.F_SOL, abi.F_RES, abi.F_CGS_RHS):
            a, b = hg.get_field(lo, f), ho.get_field(lo, f)
            print(f"level {lo} n={n} {mode:13s} field {f}: equal={np.array_equal(a, b)} maxdiff={np.abs(a - b).max():.3e} |oracle|={np.abs(b).max():.3e}")
        hg.close(); ho.close()
