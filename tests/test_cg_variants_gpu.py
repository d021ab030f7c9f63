"""The coarse-grid CG on the GPU against the oracle, in both forms (cgs_kind = CGS_CG: single reduction per iteration,
csrc/evomg_cg1.h; CGS_CG_CLASSICAL: csrc/evomg_krylov.h), through the C-ABI: same recurrence, same summation order, so the
solution and the residual field are bit-identical on every level grid from 3^2 to 129^2 — from a zero start, from a random
start, under an iteration cap, and in the folded form (COPY + SET_ZERO inside the launch)."""
import dataclasses

import numpy as np
import pytest

from evostencils_amd import abi, native, problems
from oracle import oracle

from test_cg_variants import coarse_solve, coarse_solve_3d, random_rhs, random_rhs_3d

pytestmark = pytest.mark.gpu
KINDS = {"single-reduction": abi.CGS_CG, "classical": abi.CGS_CG_CLASSICAL}


@pytest.mark.parametrize("kind", sorted(KINDS))
@pytest.mark.parametrize("lo", [1, 2, 3, 4, 5, 6, 7])
def test_coarse_solve_bit_identical_to_the_oracle(lo, kind):
    rhs = random_rhs(lo)
    xg, rg = coarse_solve(native.create, lo, KINDS[kind], rhs)
    xo, ro = coarse_solve(oracle.create, lo, KINDS[kind], rhs)
    assert np.array_equal(xg, xo) and np.array_equal(rg[1:-1, 1:-1], ro[1:-1, 1:-1])
    assert np.abs(xg).max() > 0


@pytest.mark.parametrize("kind", sorted(KINDS))
@pytest.mark.parametrize("lo", [2, 4, 5, 6, 7])
def test_nonzero_start_and_iteration_cap(lo, kind):
    rhs = random_rhs(lo, 2)
    n = rhs.shape[0]
    x0 = np.zeros((n, n))
    x0[1:-1, 1:-1] = np.random.default_rng(11).standard_normal((n - 2, n - 2))
    for kw in ({}, {"cgs_max_iterations": 1}, {"cgs_max_iterations": 5}):
        xg, rg = coarse_solve(native.create, lo, KINDS[kind], rhs, x0=x0, **kw)
        xo, ro = coarse_solve(oracle.create, lo, KINDS[kind], rhs, x0=x0, **kw)
        assert np.array_equal(xg, xo) and np.array_equal(rg[1:-1, 1:-1], ro[1:-1, 1:-1]), kw


@pytest.mark.parametrize("kind", sorted(KINDS))
@pytest.mark.parametrize("lo", [3, 5, 6, 7])
def test_folded_copy_zero_solve(lo, kind):
    """`gen_rhs = RHS; gen_error = 0; gen_mgCycle@min()` (exastencils.py:890-896) runs as ONE launch; same bits as the oracle's
    three statements, and the copy itself is written."""
    rhs = random_rhs(lo, 3)
    pd = dataclasses.replace(problems.poisson_2d(lo, lo + 1), cgs_kind=KINDS[kind])
    ops = [abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS), abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR),
           abi.make_op(abi.OP_COARSE_SOLVE, lo)]
    out = []
    for create in (native.create, oracle.create):
        h = create(pd)
        h.set_field(lo, abi.F_RHS, rhs)
        h.run_ops(ops)
        out.append([h.get_field(lo, f) for f in (abi.F_CORR, abi.F_RES, abi.F_CGS_RHS)])
    for a, b in zip(*out):
        assert np.array_equal(a[1:-1, 1:-1], b[1:-1, 1:-1])
    assert np.array_equal(out[0][2], rhs)


@pytest.mark.parametrize("kind", sorted(KINDS))
def test_coarse_solve_on_33_cubed_bit_identical_to_the_oracle(kind):
    """The coarsest grid of BASELINE config C4 on eight workgroups (evomg_cg1_slab3d / evomg_cg_multi_wg3d): solution and
    residual bit-identical to the oracle, also under an iteration cap; many solves in a row keep their exchange words apart."""
    lo = 5
    rhs = random_rhs_3d(lo, 1)
    for kw in ({}, {"cgs_max_iterations": 1}, {"cgs_max_iterations": 7}):
        xg, rg = coarse_solve_3d(native.create, lo, KINDS[kind], rhs, **kw)
        xo, ro = coarse_solve_3d(oracle.create, lo, KINDS[kind], rhs, **kw)
        assert np.array_equal(xg, xo) and np.array_equal(rg[1:-1, 1:-1, 1:-1], ro[1:-1, 1:-1, 1:-1]), kw
    h = native.create(dataclasses.replace(problems.poisson_3d(lo, lo + 1), cgs_kind=KINDS[kind]))
    h.set_field(lo, abi.F_CGS_RHS, rhs)
    ops = [abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR), abi.make_op(abi.OP_COARSE_SOLVE, lo)]
    h.run_ops(ops * 40)
    assert np.array_equal(h.get_field(lo, abi.F_CORR), xo if not kw else coarse_solve_3d(oracle.create, lo, KINDS[kind], rhs)[0])


def test_zero_right_hand_side():
    for kind in KINDS.values():
        x, r = coarse_solve(native.create, 7, kind, np.zeros((129, 129)))
        assert not x.any() and not r.any()


def test_iteration_counts_of_the_two_forms_agree_on_the_device():
    """The forms are the same Krylov process: inside a two-grid solve whose coarsest grid is the headline's (129^2) they
    take the same number of coarse-solver iterations to within a step per solve, and the solve's residual history agrees to rounding."""
    import cycles
    lo, hi = 7, 8
    smooth = lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK)
    ops = [smooth(hi), abi.make_op(abi.OP_RESIDUAL, hi, abi.F_RES, abi.F_SOL, abi.F_RHS), abi.make_op(abi.OP_RESTRICT, lo, abi.F_RHS, abi.F_RES),
           abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS), abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR), abi.make_op(abi.OP_COARSE_SOLVE, lo),
           abi.make_op(abi.OP_PROLONG_ADD, hi, abi.F_SOL, abi.F_CORR), smooth(hi)]
    its, hist = [], []
    for kind in KINDS.values():
        pd = dataclasses.replace(problems.poisson_2d(lo, hi), cgs_kind=kind)
        h = native.create(pd)
        out = h.evaluate(ops, history_capacity=64)
        its.append(h.coarse_iterations())
        hist.append(np.array(out.history))
    # (a solve's worth of coarse solves: a handful of them stop one step earlier or later in one form than in the other)
    assert abs(its[0] - its[1]) <= 0.02 * its[1] and its[0] > 300
    # each coarse solve is exact to 1e-12 of ITS right-hand side in either form: the histories part ways at that level,
    # i.e. far below 1e-10 of the initial residual
    assert len(hist[0]) == len(hist[1]) and np.abs(hist[0] - hist[1]).max() <= 1e-10 * hist[1][0]
    assert np.allclose(hist[0], hist[1], rtol=1e-3, atol=0)


def test_many_solves_in_a_row_keep_their_exchange_words_apart():
    """The eight workgroups of the 129^2 solve find each other through sequence-numbered words that are never cleared
    between launches: sixty solves back to back, every one bit-identical to the first."""
    lo = 7
    rhs = random_rhs(lo, 6)
    pd = problems.poisson_2d(lo, lo + 1)
    h = native.create(pd)
    h.set_field(lo, abi.F_CGS_RHS, rhs)
    ops = [abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR), abi.make_op(abi.OP_COARSE_SOLVE, lo)]
    h.run_ops(ops)
    first = h.get_field(lo, abi.F_CORR)
    h.run_ops(ops * 60)
    assert np.array_equal(h.get_field(lo, abi.F_CORR), first)
