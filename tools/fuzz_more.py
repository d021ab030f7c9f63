"""Developer tool: more seeds of tests/test_fuzz_gpu.py's random op programs than the suite runs, real (Poisson) and complex
(Helmholtz with its Robin columns) 2D problems, HIP backend vs CPU oracle bit for bit on every field of every level.
usage: python tools/fuzz_more.py [first_seed] [count] [3d | 3dbig]   (3d: random programs on 3D Poisson hierarchies instead;
3dbig: hierarchies that end on 129^3 and 257^3, where the marching tiles with their launch geometry run by default)"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "4")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import numpy as np
from evostencils_amd import abi, native, problems
from oracle import oracle
from test_fuzz_gpu import random_program, with_semantics

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
three_d = len(sys.argv) > 3 and sys.argv[3] in ("3d", "3dbig")
big = len(sys.argv) > 3 and sys.argv[3] == "3dbig"
bad = 0
for seed in range(first, first + count):
    for kind, lo, hi in ((("poisson3d", 5, 7), ("poisson3d", 6, 8)) if big else (("poisson3d", 2, 5), ("poisson3d", 1, 4), ("poisson3d", 3, 6)) if three_d else
                         (("poisson", 3, 7), ("helmholtz", 3, 7), ("helmholtz", 4, 8), ("poisson", 4, 8))):
        for semantics in ("textbook", "exastencils-v1.1"):
            if kind == "poisson3d":
                pd = problems.poisson_3d(lo, hi)
            elif kind == "poisson":
                pd = problems.poisson_2d(lo, hi)
            else:
                pd = problems.helmholtz_2d(lo, hi, k=0.625 * (1 << hi))
            pd = with_semantics(pd, semantics)
            rng = np.random.default_rng(31 * seed + lo)
            ops = random_program(rng, 3 if three_d else 2, lo, hi, 28)
            hg, ho = native.create(pd), oracle.create(pd)
            cplx = kind == "helmholtz"
            for L in range(lo, hi + 1):
                shape = hg.field_shape(L)
                for f in (abi.F_SOL, abi.F_RHS, abi.F_RES):
                    a = rng.standard_normal(shape) + (1j * rng.standard_normal(shape) if cplx else 0.0)
                    hg.set_field(L, f, a)
                    ho.set_field(L, f, a)
            hg.run_ops(ops)
            ho.run_ops(ops)
            for L in range(lo, hi + 1):
                for f in (abi.F_SOL, abi.F_RHS, abi.F_RES):
                    got, want = hg.get_field(L, f), ho.get_field(L, f)
                    if not np.array_equal(got, want, equal_nan=True):
                        bad += 1
                        print("MISMATCH", kind, lo, hi, semantics, "seed", seed, "level", L, "field", f, np.nanmax(np.abs(got - want)), flush=True)
            hg.close()
            ho.close()
    print("seed", seed, "done, mismatches so far", bad, flush=True)
print("total mismatches", bad)
sys.exit(1 if bad else 0)
