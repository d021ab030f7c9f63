"""Developer tool: how many outer BiCGStab iterations does the Helmholtz problem need when the preconditioner is
`reps` reference V(pre,post) cycles back to back (reps -> infinity = exact inverse of the shifted operator M)?
usage: python tools/helmholtz_multi_vcycle.py <hi> <cap> [pre post omega_index]"""
import dataclasses, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
from evostencils_amd import abi, native, problems
from search_helmholtz import cycle_ops, OMEGAS

hi, cap = int(sys.argv[1]), int(sys.argv[2])
pre, post, w = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (2, 1, 10)
lo = hi - 4
pd = dataclasses.replace(problems.helmholtz_2d(lo, hi, k=0.625 * (1 << hi)), max_iterations=cap)
h = native.create(pd, jacobi_semantics="exastencils-v1.1")
spec = [{"pre": [w] * pre, "post": [w] * post, "cgc": 18} for _ in range(hi - lo)]
for reps in (1, 2, 3, 5, 8):
    out = h.evaluate(cycle_ops(lo, hi, spec) * reps, 1, 200000.0, 0)
    print(f"hi {hi} k {pd.parameters['k']} V({pre},{post}) w={OMEGAS[w]:.2f} x{reps}: status {out.status} iterations {out.n_iterations:.0f} "
          f"time {out.time_ms:.0f} ms res/res0 {out.final_res_norm / out.init_res_norm:.2e}", flush=True)
