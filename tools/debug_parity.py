"""Developer tool: run a candidate op by op on the HIP backend and the CPU oracle and report where fields
start to differ (run on the GPU box)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import numpy as np
from evostencils_amd import abi, native, problems
from oracle import oracle
import cycles
from helpers import ops_for_case

name = sys.argv[1] if len(sys.argv) > 1 else "poisson2d_L8_v11_mixed"
problem, lo, hi, _ = cycles.CASES[name]
gen, ops = ops_for_case(name)
pd = problems.BY_NAME[problem](lo, hi)
hg, ho = native.create(pd), oracle.create(pd)
fields = [abi.F_SOL, abi.F_RHS, abi.F_RES]
for it in range(3):
    for i, op in enumerate(ops):
        hg.run_ops([op]); ho.run_ops([op])
        worst = 0.0
        for L in range(lo, hi + 1):
            for f in fields + ([abi.F_CGS_RHS] if L == lo else []):
                a, b = hg.get_field(L, f), ho.get_field(L, f)
                d = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
                worst = max(worst, d)
        print(f"it {it} op {i:2d} {op!r}: worst rel diff {worst:.3e}")
og = hg.evaluate(ops, 1, 0, 128); oo = ho.evaluate(ops, 1, 0, 128)
print("iters", og.n_iterations, oo.n_iterations)
print("hist rel", np.abs(og.history - oo.history) / oo.history)
