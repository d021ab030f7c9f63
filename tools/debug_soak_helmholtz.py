"""Developer tool: the Helmholtz part of tools/soak_candidates.py with the size of the differences shown — per candidate the
first outer iteration whose norm differs from the oracle's and how the relative difference grows (a preconditioner that does not
contract amplifies the last-bit differences of the coarse BiCGStab from iteration to iteration; a bug shows up at once).
usage: python tools/debug_soak_helmholtz.py [n] [seed] [lo hi]"""
import os, sys, dataclasses
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np
from evostencils_amd import candidates, native, problems
from evostencils_amd.code_generation.hip import ProgramGenerator
from helpers import context_for
from oracle import oracle

nh = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4243
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3, 6)
pdh = dataclasses.replace(problems.helmholtz_2d(lo, hi, k=20.0), max_iterations=40)
gen = ProgramGenerator(problem=pdh, library=native.library())
gen.initialize_code_generation(lo, hi)
pop = candidates.population(lambda: context_for(gen, lo, hi), hi - lo, 2, nh, seed=seed)
ref = oracle.create(gen.problem)        # the generator's descriptor: its `with jacobi` semantics (default exastencils-v1.1), not the bare problem's
for k, expression in enumerate(pop):
    ops = gen.generate_op_program(expression, lo, hi)
    got = gen._ensure_handle(lo, hi, None).evaluate(ops, 1, 0, 64)
    want = ref.evaluate(ops, 1, 0, 64)
    m = min(len(got.history), len(want.history))
    g, w = got.history[:m], want.history[:m]
    rel = np.abs(g - w) / np.maximum(np.abs(w), 1e-300)
    first = int(np.argmax(rel > 0)) if np.any(rel > 0) else -1
    print(k, "iterations", got.n_iterations, want.n_iterations, "first differing norm", first, "rel diff there %.2e" % (rel[first] if first >= 0 else 0.0),
          "after 5 more %.2e" % (rel[min(first + 5, m - 1)] if first >= 0 else 0.0), "max %.2e" % rel.max(), "growth of the norm itself: %.2e -> %.2e" % (w[0], w[-1]), flush=True)
