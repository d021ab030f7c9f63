"""The coarse-grid CG in its two forms, on the oracle (CPU): cgs_kind = CGS_CG runs the single-reduction recurrence of
Chronopoulos & Gear on real 2D grids up to 129^2 (csrc/evomg_cg1.h, oracle: coarse_cg1), CGS_CG_CLASSICAL the textbook
two-reduction form on every grid. Same configuration as the reference's (example_problems/Poisson/2D_FD_Poisson_fromL2.exa3:12-14:
CG, at most 1000 iterations, reduction 1e-12), same stopping rule; the reference's one recorded number does not depend on
the coarse solver's arithmetic (oracle/APPB_MATRIX.md, row coarse=exact), so the form is ours to choose — these tests pin
that the two forms are the same solver up to rounding."""
import dataclasses

import numpy as np
import pytest

from evostencils_amd import abi, problems
from oracle import oracle


def coarse_solve(lib_create, lo, kind, rhs, x0=None, **kw):
    pd = dataclasses.replace(problems.poisson_2d(lo, lo + 1), cgs_kind=kind, **kw)
    h = lib_create(pd)
    h.set_field(lo, abi.F_CGS_RHS, rhs)
    ops = [abi.make_op(abi.OP_COARSE_SOLVE, lo)]
    if x0 is None:
        ops.insert(0, abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR))
    else:
        h.set_field(lo, abi.F_CORR, x0)
    h.run_ops(ops)
    return h.get_field(lo, abi.F_CORR), h.get_field(lo, abi.F_RES)


def random_rhs(lo, seed=0):
    n = (1 << lo) + 1
    rhs = np.zeros((n, n))
    rhs[1:-1, 1:-1] = np.random.default_rng(100 * lo + seed).standard_normal((n - 2, n - 2))
    return rhs


def poisson_apply(u):
    n = u.shape[0]
    h2 = (1.0 / (n - 1)) ** 2
    v = np.zeros_like(u)
    v[1:-1, 1:-1] = (4 * u[1:-1, 1:-1] - u[:-2, 1:-1] - u[2:, 1:-1] - u[1:-1, :-2] - u[1:-1, 2:]) / h2
    return v


@pytest.mark.parametrize("lo", [1, 2, 3, 4, 5, 6, 7])
def test_single_reduction_form_solves_what_the_classical_form_solves(lo):
    rhs = random_rhs(lo)
    x1, r1 = coarse_solve(oracle.create, lo, abi.CGS_CG, rhs)
    x0, r0 = coarse_solve(oracle.create, lo, abi.CGS_CG_CLASSICAL, rhs)
    scale = np.abs(x0).max()
    assert np.abs(x1 - x0).max() <= 1e-10 * scale
    # both leave the recursively updated residual below the tolerance, and it is the true residual to rounding
    for x, r in ((x1, r1), (x0, r0)):
        assert np.linalg.norm(r[1:-1, 1:-1]) <= 1e-12 * np.linalg.norm(rhs)
        true = rhs - poisson_apply(x)
        assert np.linalg.norm(true[1:-1, 1:-1] - r[1:-1, 1:-1]) <= 1e-10 * np.linalg.norm(rhs)


@pytest.mark.parametrize("lo", [3, 5, 7])
def test_single_reduction_form_from_a_nonzero_start_and_under_the_iteration_cap(lo):
    rhs = random_rhs(lo, 1)
    n = rhs.shape[0]
    x0 = np.zeros((n, n))
    x0[1:-1, 1:-1] = np.random.default_rng(7).standard_normal((n - 2, n - 2))
    a, _ = coarse_solve(oracle.create, lo, abi.CGS_CG, rhs, x0=x0)
    b, _ = coarse_solve(oracle.create, lo, abi.CGS_CG_CLASSICAL, rhs, x0=x0)
    assert np.abs(a - b).max() <= 1e-10 * np.abs(b).max()
    # a cap of three iterations: both forms have taken exactly three steps of the same Krylov process
    a, ra = coarse_solve(oracle.create, lo, abi.CGS_CG, rhs, cgs_max_iterations=3)
    b, rb = coarse_solve(oracle.create, lo, abi.CGS_CG_CLASSICAL, rhs, cgs_max_iterations=3)
    assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
    assert np.abs(ra - rb).max() <= 1e-10 * np.abs(rb).max()


def test_zero_right_hand_side_is_left_alone():
    for kind in (abi.CGS_CG, abi.CGS_CG_CLASSICAL):
        x, r = coarse_solve(oracle.create, 4, kind, np.zeros((17, 17)))
        assert not x.any() and not r.any()


def coarse_solve_3d(lib_create, lo, kind, rhs, **kw):
    h = lib_create(dataclasses.replace(problems.poisson_3d(lo, lo + 1), cgs_kind=kind, **kw))
    h.set_field(lo, abi.F_CGS_RHS, rhs)
    h.run_ops([abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR), abi.make_op(abi.OP_COARSE_SOLVE, lo)])
    return h.get_field(lo, abi.F_CORR), h.get_field(lo, abi.F_RES)


def random_rhs_3d(lo, seed=0):
    n = (1 << lo) + 1
    rhs = np.zeros((n, n, n))
    rhs[1:-1, 1:-1, 1:-1] = np.random.default_rng(300 * lo + seed).standard_normal((n - 2,) * 3)
    return rhs


def test_grids_without_a_single_reduction_kernel_keep_the_classical_form():
    """cgs_kind = CGS_CG on grids the single-reduction form does not serve (3D other than 33^3, 2D beyond 129^2) is the
    classical recurrence, bit for bit."""
    for lo in (2, 3, 4):
        rhs = random_rhs_3d(lo)
        a, _ = coarse_solve_3d(oracle.create, lo, abi.CGS_CG, rhs)
        b, _ = coarse_solve_3d(oracle.create, lo, abi.CGS_CG_CLASSICAL, rhs)
        assert np.array_equal(a, b)
    rhs = random_rhs(8)
    a, _ = coarse_solve(oracle.create, 8, abi.CGS_CG, rhs, cgs_max_iterations=20)
    b, _ = coarse_solve(oracle.create, 8, abi.CGS_CG_CLASSICAL, rhs, cgs_max_iterations=20)
    assert np.array_equal(a, b)


def test_single_reduction_form_on_the_3d_grid_of_eight_slabs():
    """33^3 (the coarsest grid of BASELINE config C4): the two forms solve the same system to the same tolerance."""
    lo = 5
    rhs = random_rhs_3d(lo)
    x1, r1 = coarse_solve_3d(oracle.create, lo, abi.CGS_CG, rhs)
    x0, r0 = coarse_solve_3d(oracle.create, lo, abi.CGS_CG_CLASSICAL, rhs)
    assert not np.array_equal(x1, x0)
    assert np.abs(x1 - x0).max() <= 1e-10 * np.abs(x0).max()
    for r in (r1, r0):
        assert np.linalg.norm(r[1:-1, 1:-1, 1:-1]) <= 1e-12 * np.linalg.norm(rhs)


def test_unknown_solver_kind_is_rejected():
    pd = dataclasses.replace(problems.poisson_2d(3, 4), cgs_kind=7)
    with pytest.raises(abi.BackendError):
        oracle.create(pd)
