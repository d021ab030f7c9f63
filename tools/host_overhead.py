"""Developer tool: where the wall time of one candidate evaluation goes besides the solve itself."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import bench
from evostencils_amd import native, problems
from evostencils_amd.code_generation.hip import ProgramGenerator
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7, 12)
gen = ProgramGenerator(problem=problems.poisson_2d(lo, hi), library=native.library())
gen.initialize_code_generation(lo, hi)
e = bench.build_candidate(gen, lo, hi)
ops = gen.generate_op_program(e, lo, hi)
h = gen._handle
for _ in range(2): h.evaluate(ops, 1, 0, 128)
n = 10
t0 = time.perf_counter(); s = 0.0
for _ in range(n):
    out = h.evaluate(ops, 1, 0, 128); s += out.time_ms
t1 = time.perf_counter()
print("evaluate wall %.3f ms, solve timer %.3f ms, difference %.3f ms" % ((t1 - t0) / n * 1e3, s / n, (t1 - t0) / n * 1e3 - s / n))
t0 = time.perf_counter()
for _ in range(n): gen.generate_op_program(e, lo, hi)
print("lowering %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
t0 = time.perf_counter()
for _ in range(n): gen.generate_and_evaluate(e, None, lo, hi, "", evaluation_samples=1)
print("generate_and_evaluate wall %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
