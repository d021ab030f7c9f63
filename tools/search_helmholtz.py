"""Developer tool: random search for a V-type Helmholtz preconditioner cycle (red-black Gauss-Seidel sweeps only — under
the reference pipeline's semantics `with jacobi` sweeps are no-ops) that lets the outer BiCGStab converge at k h = 0.625 on
fine grids, where the hand-written reference V(2,1) cycle (Helmholtz exa3:202-272) does not within its 10000 iterations.
Relaxation factors come from the grammar's set np.linspace(0.1, 1.9, 37), so every candidate is producible by the grammar.
usage: python tools/search_helmholtz.py <hi> <n_candidates> <cap> [seed]   -> prints one line per candidate + the best"""
import dataclasses, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import numpy as np
from evostencils_amd import abi, native, problems

OMEGAS = np.linspace(0.1, 1.9, 37)


def cycle_ops(lo, hi, spec):
    """spec: per level hi..lo+1 (finest first) a dict pre=[omega idx...], post=[...], cgc=omega idx"""
    one = (1, 1, 1)
    ops = []
    for d, L in enumerate(range(hi, lo, -1)):
        for w in spec[d]["pre"]:
            ops.append(abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, OMEGAS[w]))
        ops.append(abi.make_op(abi.OP_RESIDUAL, L, abi.F_RES, abi.F_SOL, abi.F_RHS))
        ops.append(abi.make_op(abi.OP_RESTRICT, L - 1, abi.F_RHS, abi.F_RES))
        if L - 1 > lo:
            ops.append(abi.make_op(abi.OP_SET_ZERO, L - 1, abi.F_SOL))
    ops.append(abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS))
    ops.append(abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR))
    ops.append(abi.make_op(abi.OP_COARSE_SOLVE, lo))
    for L in range(lo + 1, hi + 1):
        d = hi - L
        ops.append(abi.make_op(abi.OP_PROLONG_ADD, L, abi.F_SOL, abi.F_SOL, omega=OMEGAS[spec[d]["cgc"]]))
        for w in spec[d]["post"]:
            ops.append(abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, OMEGAS[w]))
    return ops


def random_spec(rng, depth):
    spec = []
    for _ in range(depth):
        spec.append({"pre": [int(rng.integers(4, 23)) for _ in range(int(rng.integers(0, 4)))],
                     "post": [int(rng.integers(4, 23)) for _ in range(int(rng.integers(0, 4)))],
                     "cgc": int(rng.integers(2, 25))})
    return spec


REFERENCE = lambda depth: [{"pre": [10, 10], "post": [10], "cgc": 18} for _ in range(depth)]     # w = 0.6, cgc 1.0


def main():
    hi, count, cap = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    specs_file = sys.argv[5] if len(sys.argv) > 5 else None
    lo = hi - 4
    pd = dataclasses.replace(problems.helmholtz_2d(lo, hi, k=0.625 * (1 << hi)), max_iterations=cap)
    h = native.create(pd, jacobi_semantics="exastencils-v1.1")
    rng = np.random.default_rng(seed)
    if specs_file:
        specs = [json.loads(l)["spec"] for l in open(specs_file) if l.strip()]
    else:
        specs = [REFERENCE(hi - lo)] + [random_spec(rng, hi - lo) for _ in range(count)]
    results = []
    for i, spec in enumerate(specs):
        out = h.evaluate(cycle_ops(lo, hi, spec), 1, 120000.0, 0)
        results.append((out.n_iterations if out.status == abi.SOLVE_CONVERGED else 10 ** 9, out.time_ms, i))
        print(json.dumps({"i": i, "hi": hi, "status": out.status, "iterations": out.n_iterations, "time_ms": out.time_ms,
                          "res": out.final_res_norm / out.init_res_norm, "spec": spec}), flush=True)
    results.sort()
    print("BEST", [(r[0], round(r[1], 1), r[2]) for r in results[:8]])


if __name__ == "__main__":
    main()
