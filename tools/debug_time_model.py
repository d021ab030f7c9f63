"""Developer tool: the bench line's population with the isolated time objective, candidate by candidate — sequential time, modelled
time, iterations, coarse iterations, and which candidates deviate most. usage: python tools/debug_time_model.py [n] [in_flight]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import numpy as np
from evostencils_amd import candidates, population
from evostencils_amd.code_generation.hip import ProgramGenerator
from evostencils_amd import problems
from helpers import context_for

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 12
LO, HI = 7, 11
make = lambda: _make()
def _make():
    g = ProgramGenerator(problem=problems.poisson_2d(LO, HI))
    g.initialize_code_generation(LO, HI)
    return g
gen = make()
pop = candidates.population(lambda: context_for(gen, LO, HI), HI - LO, 2, N, seed=2024)
evaluate = lambda g, i: g.generate_and_evaluate(pop[i], None, LO, HI, "", evaluation_samples=1)
evaluate(gen, 0)
seq, ks = [], []
for i in range(N):
    seq.append(evaluate(gen, i))
    ks.append(gen.last_coarse_iterations)
workers = population.ConcurrentEvaluator(make, W)
workers.map(evaluate, range(W))
log = {}
def time_fn(g, i, its, k):
    before = g._coarse_iteration_ms.get((LO, HI))
    t = g.isolated_time_ms(g.generate_op_program(pop[i], LO, HI), LO, HI, its, coarse_iterations=k)
    log[i] = (id(g) % 1000, before if isinstance(before, float) else None, g._coarse_iteration_ms.get((LO, HI)))
    return t
rows = population.evaluate_population_in_flight(workers, evaluate, N, time_fn=time_fn)
workers.close()
ts, tm = np.array([s[0] for s in seq]), np.array([r[2] for r in rows])
rel = np.abs(tm - ts) / ts
print("max rel deviation %.3f, median ratio %.4f" % (rel.max(), np.median(tm / ts)))
for i in np.argsort(-rel)[:12]:
    print(f"candidate {i}: sequential {ts[i]:.2f} ms, modelled {tm[i]:.2f} ms ({100 * rel[i]:.1f} %), iterations {seq[i][2]}, coarse iterations {ks[i]} "
          f"({(ks[i] or 0) / max(seq[i][2], 1):.1f} per cycle), worker {log[i][0]}, coarse cost known before {log[i][1]}, after {log[i][2]}")
settled = sorted({v[2] for v in log.values() if isinstance(v[2], float)})
print("settled coarse-iteration costs over the workers (us):", [round(1e3 * v, 3) for v in settled])
