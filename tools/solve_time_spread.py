"""Run-to-run spread of the solve time of one candidate (the time objective's noise floor): C2-like V(1,1) RB-GS cycle on
levels lo..hi, `reps` evaluations one after the other. usage: python tools/solve_time_spread.py [lo hi reps] [classical]"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import dataclasses
import numpy as np
from evostencils_amd import abi, native, problems, candidates as cand

lo, hi, reps = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (7, 11, 30)))
kind = abi.CGS_CG_CLASSICAL if "classical" in sys.argv else abi.CGS_CG
pd = dataclasses.replace(problems.poisson_2d(lo, hi), cgs_kind=kind)
h = native.create(pd)
ops = cand.v_cycle_ops(lo, hi, 1, 1)
h.evaluate(ops, 1, 0, 0)
t = np.array([h.evaluate(ops, 1, 0, 0).time_ms for _ in range(reps)])
print("%s levels %d..%d: solve %.3f ms median, min %.3f, max %.3f, (max-min)/median %.3f, std/median %.4f; coarse iterations %d" % (
    "classical" if kind == abi.CGS_CG_CLASSICAL else "single-reduction", lo, hi, np.median(t), t.min(), t.max(), (t.max() - t.min()) / np.median(t),
    t.std() / np.median(t), h.coarse_iterations()))
