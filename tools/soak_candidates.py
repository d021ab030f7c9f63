"""Developer tool: soak test — many seeded random candidates (2D and 3D), HIP backend vs CPU oracle: identical
iteration counts, convergence factors within 1e-10. usage: python tools/soak_candidates.py [n2d] [n3d] [seed]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
os.environ.setdefault("OMP_NUM_THREADS", "8")
from evostencils_amd import candidates, native, problems
from evostencils_amd.code_generation.hip import ProgramGenerator
from helpers import context_for, generator_for
from oracle import oracle

n2d = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n3d = int(sys.argv[2]) if len(sys.argv) > 2 else 10
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 99
bad = 0
for name, make, lo, hi, n, dim in (("2D_FD_Poisson_fromL2", problems.poisson_2d, 4, 8, n2d, 2),
                                   ("2D_FD_Poisson_fromL2", problems.poisson_2d, 5, 9, n2d // 2, 2),
                                   ("3D_FD_Poisson_fromL2", problems.poisson_3d, 5, 6, n3d, 3)):
    gen = generator_for(name, lo, hi, library=native.library())
    gen.initialize_code_generation(lo, hi)
    pop = candidates.population(lambda: context_for(gen, lo, hi), hi - lo, dim, n, seed=seed)
    ref = oracle.create(make(lo, hi))
    for k, expression in enumerate(pop):
        ops = gen.generate_op_program(expression, lo, hi)
        t, rho, its = gen.generate_and_evaluate(expression, None, lo, hi, "", evaluation_samples=1)
        rt, rrho, rits = ProgramGenerator.interpret_outcome(ref.evaluate(ops, 1, 0, 128), 1e100)
        ok = its == rits and abs(rho - rrho) <= 1e-10 * abs(rrho)
        if not ok:
            bad += 1
            print("MISMATCH", name, lo, hi, k, (rho, its), (rrho, rits), flush=True)
    print(name, lo, hi, "checked", n, flush=True)
    gen.close()
# Helmholtz (complex, Robin, MG-preconditioned BiCGStab outer solver): capped outer iterations, histories compared
import dataclasses
import numpy as np
nh = int(sys.argv[4]) if len(sys.argv) > 4 else 10
lo, hi = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (3, 6)
pdh = dataclasses.replace(problems.helmholtz_2d(lo, hi, k=20.0), max_iterations=40)
gen = ProgramGenerator(problem=pdh, library=native.library())
gen.initialize_code_generation(lo, hi)
pop = candidates.population(lambda: context_for(gen, lo, hi), hi - lo, 2, nh, seed=seed + 1)
ref = oracle.create(gen.problem)        # the generator's descriptor: its `with jacobi` semantics (default exastencils-v1.1), not the bare problem's
for k, expression in enumerate(pop):
    ops = gen.generate_op_program(expression, lo, hi)
    got = gen._ensure_handle(lo, hi, None).evaluate(ops, 1, 0, 64)
    want = ref.evaluate(ops, 1, 0, 64)
    m = min(len(got.history), len(want.history))
    finite = np.isfinite(want.history[:m]) & (want.history[:m] < 1e200)
    ok = got.n_iterations == want.n_iterations and got.status == want.status and \
        np.all(np.abs(got.history[:m][finite] - want.history[:m][finite]) <= 1e-6 * np.abs(want.history[:m][finite]))
    if not ok:
        bad += 1
        print("MISMATCH helmholtz", k, got.n_iterations, want.n_iterations, got.status, want.status, flush=True)
print("helmholtz", lo, hi, "checked", nh, flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
