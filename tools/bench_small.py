"""Back-to-back time of one point-smoother launch on small 2D levels (bench_op: HIP events around `reps` launches).
usage: python tools/bench_small.py [poisson|helmholtz]"""
import sys
sys.path[:0] = ['/root/repo', '/root/repo/tests']
from evostencils_amd import abi, native, problems
kind = sys.argv[1] if len(sys.argv) > 1 else "helmholtz"
lo, hi = 3, 8
pd = problems.poisson_2d(lo, hi) if kind == "poisson" else problems.helmholtz_2d(lo, hi, k=0.625 * (1 << hi))
h = native.create(pd)
for L in range(4, 9):
    out = []
    for part, name in ((abi.PART_RED_BLACK, "rbgs"), (abi.PART_SINGLE, "jacobi")):
        op = abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, part, (1, 1, 1), 0.8)
        b = h.bench_op(op, 10, 200)
        out.append("%s %.2f us" % (name, b.ms * 1e3))
    print("%s %d^2: %s" % (kind, (1 << L) + 1, ", ".join(out)), flush=True)
h.close()
