"""One BiCGStab coarse solve of the Helmholtz problem at 33^2 .. 257^2: single workgroup vs launch-per-op (EVOMG_KB_MIN_N).
usage: python tools/coarse_bicgstab_sizes.py"""
import os, sys, time
sys.path[:0] = ['/root/repo', '/root/repo/tests']
import numpy as np
from evostencils_amd import abi, native, problems

for lo in (5, 6, 7, 8):
    n = (1 << lo) + 1
    rng = np.random.default_rng(lo)
    rhs = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    line = []
    for min_n in ("100000", "33"):
        os.environ["EVOMG_KB_MIN_N"] = min_n
        pd = problems.helmholtz_2d(lo, lo + 1, k=0.625 * (1 << (lo + 1)))
        h = native.create(pd)
        h.set_field(lo, abi.F_CGS_RHS, rhs)
        ops = [abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR), abi.make_op(abi.OP_COARSE_SOLVE, lo)] * 20
        h.run_ops(ops)
        h.get_field(lo, abi.F_CORR)
        t0 = time.perf_counter()
        h.run_ops(ops)
        h.get_field(lo, abi.F_CORR)
        dt = (time.perf_counter() - t0) / 20
        line.append("%s %.1f us" % ("one workgroup" if min_n != "33" else "per launch", dt * 1e6))
        h.close()
    print("%d^2:" % n, ", ".join(line), flush=True)
