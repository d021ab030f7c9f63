"""Developer tool: run one of the BASELINE.json configurations once on the GPU (for rocprofv3).
usage: python tools/run_config.py c4|c2|c3|c5|t9|t8|fas|ela|p3d|h7"""
import os, sys, dataclasses
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
from evostencils_amd import abi, candidates, native, problems
vcycle = candidates.v_cycle_ops
reference_vcycle_ops = lambda lo, hi: candidates.v_cycle_ops(lo, hi, 2, 1, omega=0.6)

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
if cfg == "c4":
    lo, hi = 5, 9
    pd = problems.poisson_3d(lo, hi)
    ops = vcycle(lo, hi, 2, 2, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 1, 1), 0.8))
elif cfg == "c2":
    lo, hi = 7, 12
    pd = problems.poisson_2d(lo, hi)
    ops = vcycle(lo, hi, 1, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.0))
elif cfg == "c5":
    lo, hi = 7, 11
    pd = problems.poisson_2d(lo, hi)
    ops = vcycle(lo, hi, 1, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.0))
elif cfg == "fas":
    # the reference's non-linear FAS example at its own level range (6..10), V(1,1) Newton-Jacobi
    import cycles
    from test_fas_lowering import fas_context, fas_generator
    gen = fas_generator(levels=(6, 10))
    sm = [[cycles._nw(14)] for _ in range(4)]
    ops = gen.generate_op_program(cycles.fas_v_cycle(fas_context(gen, 4), 4, sm, sm)[0])
    pd, lo, hi = gen.problem_description(), 6, 10
elif cfg == "ela":
    # the reference's linear-elasticity example at its own level range (4..8, 257^2, two unknowns): the problem's own
    # solver settings as a candidate — V(2,1), collective red-black w = 1.25
    import cycles
    from helpers import build_case, context_for, generator_for
    lo, hi = 4, 8
    gen = generator_for("2D_FD_LinearElasticity_fromL2", lo, hi)
    rb = [("rbgs", 23)]
    expression, _ = cycles.v_cycle(context_for(gen, lo, hi), hi - lo, [rb * 2] * (hi - lo), [rb] * (hi - lo))
    ops = gen.generate_op_program(expression, lo, hi)
    pd = gen.problem
elif cfg == "p3d":
    # the reference's 3D Poisson example at its own level range (2..6, 65^3): its solver settings, V(2,1) red-black w = 1.25
    lo, hi = 2, 6
    pd = problems.poisson_3d(lo, hi)
    ops = vcycle(lo, hi, 2, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.25))
elif cfg == "h7":
    # the reference's Helmholtz example at its own level range (3..7, 129^2, k = 80): its hand-written V(2,1) cycle
    lo, hi = 3, 7
    pd = problems.helmholtz_2d(lo, hi, k=80.0)
    ops = reference_vcycle_ops(lo, hi)
elif cfg in ("t9", "t8"):
    # the reference's own operating size (notebooks/tutorial.ipynb: levels 5..9 = 513^2, five levels)
    lo, hi = (5, 9) if cfg == "t9" else (4, 8)
    pd = problems.poisson_2d(lo, hi)
    ops = vcycle(lo, hi, 1, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.0))
else:
    lo, hi = 7, 11
    pd = dataclasses.replace(problems.helmholtz_2d(lo, hi, k=0.625 * (1 << hi)), max_iterations=50)
    ops = reference_vcycle_ops(lo, hi)
h = native.create(pd)
out = h.evaluate(ops, 1, 0, 0)
out = h.evaluate(ops, 1, 0, 0)
best = min(h.evaluate(ops, 1, 0, 0).time_ms for _ in range(int(os.environ.get("RUN_CONFIG_REPEATS", "0")))) if os.environ.get("RUN_CONFIG_REPEATS") else None
if best is not None:
    print(cfg, "best of repeats %.3f ms" % best)
print(cfg, "iterations", out.n_iterations, "rho %.4f" % out.conv_factor, "time %.2f ms" % out.time_ms, "status", out.status)
