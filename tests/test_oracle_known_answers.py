"""Pins the CPU oracle independently of ExaStencils (which cannot run in this pipeline) with the
known-answer tests the reference's problem files imply:

 * 2D Poisson: u = cos(pi x) - sin(2 pi y) is the boundary condition AND satisfies the PDE
   (example_problems/Poisson/2D_FD_Poisson_fromL2.exa2:4-7) -> converged discrete solution has an O(h^2) error.
 * 3D Poisson: the boundary data x^2 - y^2/2 - z^2/2 is harmonic and quadratic, rhs = 0
   (Poisson/3D_FD_Poisson_fromL2.exa2:6-9) -> the 7-point stencil reproduces it exactly.
 * textbook convergence factors: V(1,1) RB-GS + FW + bilinear: rho < 0.1;  V(1,1) Jacobi w=0.8: rho ~ 0.36.
"""
import numpy as np
import pytest

from evostencils_amd import abi, problems
from oracle import oracle
from helpers import ops_for_case
import cycles


def simple_vcycle(lo, hi, pre=1, post=1, om=1.0, part=abi.PART_RED_BLACK, block=(1, 1, 1)):
    ops = []
    for L in range(hi, lo, -1):
        for _ in range(pre):
            ops.append(abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, part, block, om))
        ops.append(abi.make_op(abi.OP_RESIDUAL, L, abi.F_RES, abi.F_SOL, abi.F_RHS))
        ops.append(abi.make_op(abi.OP_RESTRICT, L - 1, abi.F_RHS, abi.F_RES))
        if L - 1 > lo:
            ops.append(abi.make_op(abi.OP_SET_ZERO, L - 1, abi.F_SOL))
    ops.append(abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS))
    ops.append(abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR))
    ops.append(abi.make_op(abi.OP_COARSE_SOLVE, lo))
    for L in range(lo + 1, hi + 1):
        ops.append(abi.make_op(abi.OP_PROLONG_ADD, L, abi.F_SOL, abi.F_SOL, omega=1.0))
        for _ in range(post):
            ops.append(abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, part, block, om))
    return ops


def exact_2d(level):
    n = (1 << level) + 1
    x = np.arange(n) / (n - 1)
    return np.cos(np.pi * x[None, :]) - np.sin(2 * np.pi * x[:, None])


def test_poisson2d_discretisation_error_is_second_order():
    errs = []
    for hi in (5, 6, 7):
        h = oracle.create(problems.poisson_2d(3, hi))
        out = h.evaluate(simple_vcycle(3, hi), 1, 0, 0)
        assert out.status == abi.SOLVE_CONVERGED
        errs.append(np.abs(h.get_field(hi, abi.F_SOL) - exact_2d(hi)).max())
    assert 3.7 < errs[0] / errs[1] < 4.3 and 3.7 < errs[1] / errs[2] < 4.3


def test_poisson3d_quadratic_solution_is_reproduced_exactly():
    hi = 4
    h = oracle.create(problems.poisson_3d(2, hi))
    out = h.evaluate(simple_vcycle(2, hi, pre=2, post=2, om=0.8, part=abi.PART_SINGLE), 1, 0, 0)
    assert out.status == abi.SOLVE_CONVERGED
    n = (1 << hi) + 1
    x = np.arange(n) / (n - 1)
    exact = x[None, None, :] ** 2 - 0.5 * x[None, :, None] ** 2 - 0.5 * x[:, None, None] ** 2
    assert np.abs(h.get_field(hi, abi.F_SOL) - exact).max() < 1e-11


def test_textbook_convergence_factors():
    h = oracle.create(problems.poisson_2d(3, 7))
    rb = h.evaluate(simple_vcycle(3, 7), 1, 0, 0)
    assert rb.status == abi.SOLVE_CONVERGED and rb.conv_factor < 0.1
    jac = h.evaluate(simple_vcycle(3, 7, om=0.8, part=abi.PART_SINGLE), 1, 0, 0)
    assert jac.status == abi.SOLVE_CONVERGED and 0.30 < jac.conv_factor < 0.40


def test_history_is_monotone_and_consistent():
    h = oracle.create(problems.poisson_2d(3, 6))
    out = h.evaluate(simple_vcycle(3, 6), 1, 0, 128)
    hist = out.history
    assert len(hist) == int(out.n_iterations) + 1
    assert hist[0] == out.init_res_norm and hist[-1] == out.final_res_norm
    assert np.all(hist[1:] < hist[:-1])
    assert hist[-1] <= 1e-12 * hist[0]
    assert abs(out.conv_factor - (hist[-1] / hist[0]) ** (1.0 / out.n_iterations)) < 1e-15


def test_block_tiles_cover_every_inner_node_once():
    """A block sweep with w=1 on a system whose exact solution is known must leave the exact
    solution fixed (every inner node is written by exactly one tile; truncated tiles included)."""
    pd = problems.poisson_2d(3, 5)
    h = oracle.create(pd)
    rng = np.random.default_rng(5)
    n = 33
    x = rng.standard_normal((n, n))
    # f := A x  computed through the residual op with rhs = 0:  res = 0 - A x
    h.set_field(5, abi.F_SOL, x)
    h.set_field(5, abi.F_RHS, np.zeros((n, n)))
    h.run_ops([abi.make_op(abi.OP_RESIDUAL, 5, abi.F_RES, abi.F_SOL, abi.F_RHS)])
    f = -h.get_field(5, abi.F_RES)
    f[0, :] = f[-1, :] = f[:, 0] = f[:, -1] = 0
    for block in [(2, 1, 1), (1, 2, 1), (2, 2, 1), (3, 2, 1), (1, 6, 1), (4, 2, 1), (8, 1, 1)]:
        h.set_field(5, abi.F_SOL, x)
        h.set_field(5, abi.F_RHS, f)
        h.run_ops([abi.make_op(abi.OP_SMOOTH, 5, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, block, 1.0)])
        assert np.abs(h.get_field(5, abi.F_SOL) - x).max() < 1e-9, block


def test_rejects_malformed_programs():
    h = oracle.create(problems.poisson_2d(3, 5))
    bad = [abi.make_op(abi.OP_SMOOTH, 9, abi.F_SOL, 0, abi.F_RHS)]
    rc, _ = h.try_evaluate(bad, 1, 0, 0)
    assert rc == abi.ERR_INVALID
    rc, _ = h.try_evaluate([abi.make_op(abi.OP_SMOOTH, 5, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (2, 1, 1))], 1, 0, 0)
    assert rc == abi.ERR_UNSUPPORTED
    rc, _ = h.try_evaluate([abi.make_op(abi.OP_COARSE_SOLVE, 4)], 1, 0, 0)
    assert rc == abi.ERR_INVALID


@pytest.mark.parametrize("name", sorted(cycles.CASES))
def test_golden_cases_run_on_the_oracle(name):
    """Every candidate pinned by a golden string is also executable: lowering -> op program -> oracle."""
    problem, lo, hi, _ = cycles.CASES[name]
    if name == "poisson2d_tutorial":
        lo, hi = 5, 9
    gen, ops = ops_for_case(name)
    # one block shape per unknown: only where the sweep's result is never used (tests/test_systems_oracle.py)
    semantics = "exastencils-v1.1" if name == "elasticity2d_L5_mixed_shapes" else None
    h = oracle.create(problems.BY_NAME[problem](lo, hi), jacobi_semantics=semantics)
    out = h.evaluate(ops, 1, 0, 128)
    assert out.status in (abi.SOLVE_CONVERGED, abi.SOLVE_MAXITER)
    assert np.isfinite(out.final_res_norm) and out.final_res_norm < out.init_res_norm


def test_tutorial_cycle_converges_slowly_like_the_recorded_run():
    """notebooks/tutorial.ipynb:3373 records rho = 0.9213 and 100 iterations (iteration cap) for this
    cycle on an unknown machine; the restated numerics must land in the same regime: a convergent but
    slow solver that hits the 100-iteration cap."""
    gen, ops = ops_for_case("poisson2d_tutorial")
    h = oracle.create(problems.poisson_2d(5, 9))
    out = h.evaluate(ops, 1, 0, 128)
    assert out.status == abi.SOLVE_MAXITER and out.n_iterations == 100
    assert 0.85 < out.conv_factor < 0.97


def test_host_tuned_build_of_the_oracle_gives_the_same_bits():
    """bench.py times the oracle built with -march=native on the host it runs on (oracle.native_library); same source, same
    -ffp-contract=off: the residual history of a solve is bit-identical to the portable build's."""
    import numpy as np
    from evostencils_amd import abi, problems
    from oracle import oracle
    from test_full_size_gpu import vcycle
    lo, hi = 3, 6
    ops = vcycle(lo, hi, 2, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.15))
    a = oracle.create(problems.poisson_2d(lo, hi)).evaluate(ops, 1, 0, 64)
    b = oracle.create(problems.poisson_2d(lo, hi), lib=oracle.native_library()).evaluate(ops, 1, 0, 64)
    assert a.n_iterations == b.n_iterations and np.array_equal(a.history, b.history)
