import sys
sys.path[:0]=["/root/repo","/root/repo/tests"]
from multiprocessing.pool import ThreadPool
from evostencils_amd import candidates, problems, native
from evostencils_amd.code_generation.hip import ProgramGeneratorPool, ProgramGenerator
from helpers import context_for
lo, hi, n = 7, 11, 12
def make():
    g = ProgramGenerator(problem=problems.poisson_2d(lo, hi), library=native.library())
    g.initialize_code_generation(lo, hi)
    return g
single = make()
pop = candidates.population(lambda: context_for(single, lo, hi), hi - lo, 2, n, seed=2024)
evaluate = lambda g, e: g.generate_and_evaluate(e, None, lo, hi, "", evaluation_samples=1)
evaluate(single, pop[0])
want = [evaluate(single, e) for e in pop]
want2 = [evaluate(single, e) for e in pop]
single.close()
for rep in range(3):
    pool = ProgramGeneratorPool(make, 4, time_objective="isolated")
    with ThreadPool(4) as tp:
        tp.map(lambda e: evaluate(pool, e), pop[:4])
        got = tp.map(lambda e: evaluate(pool, e), pop)
    costs = [g._coarse_iteration_ms.get((lo, hi)) for g in pool._generators]
    pool.close()
    print("rep", rep, "ratios", " ".join("%.3f" % (g[0] / w[0]) for g, w in zip(got, want)), "| seq vs seq", "%.3f" % max(abs(a[0]-b[0])/b[0] for a,b in zip(want2, want)))
    print("   worker coarse costs:", [("%.3f us" % (1e3*c) if isinstance(c, float) else [round(1e3*v,3) for v in c]) for c in costs])
