"""Developer tool: per candidate of the C5 slice — sequential time, iterations, two-point probe (slope, intercept), modelled time."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
from evostencils_amd import abi, candidates, native, problems
from evostencils_amd.code_generation.hip import ProgramGenerator
from helpers import context_for
LO, HI, N = 7, 11, 16
gen = ProgramGenerator(problem=problems.poisson_2d(LO, HI), library=native.library())
gen.initialize_code_generation(LO, HI)
pop = candidates.population(lambda: context_for(gen, LO, HI), HI - LO, 2, N, seed=2024)
gen.generate_and_evaluate(pop[0], None, LO, HI, "", evaluation_samples=1)
for i in range(N):
    ops = gen.generate_op_program(pop[i], LO, HI)
    t, rho, its = gen.generate_and_evaluate(pop[i], None, LO, HI, "", evaluation_samples=1)
    h = gen._handle
    k = gen.last_coarse_iterations
    probes = {n: h.time_iterations(ops, n) for n in (1, 2, 4, 8, 16, 32)}
    fixed, slope, cg = h.time_model(ops, 2, 6)
    print(f"cand {i:2d}: seq {t:8.2f} ms its {its:5.0f} -> {t / its:7.4f} ms/it | probes " +
          " ".join(f"{n}:{ms * d:7.2f}" for n, (ms, d) in probes.items()) + f" | K {k} fixed {fixed:6.3f} it {slope:7.4f} cg {1e3 * cg:6.3f} us model {fixed + slope * its + cg * k:8.2f}", flush=True)
