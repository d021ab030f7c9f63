"""Non-linear FAS backend on the GPU: the HIP kernels behind the EVOMG_OP_FAS_* vocabulary against the CPU
oracle, through the C-ABI. exp() comes from two different math libraries (device ocml, host libm), so the bar is the
floating-point tolerance north_star states — 1e-10 relative on residual norms — with identical iteration counts; single
statements agree to 1e-12."""
import numpy as np
import pytest

from evostencils_amd import abi

import cycles
from test_fas_lowering import fas_context, fas_generator
from test_fas_oracle import exact_solution, fas_problem, ops_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()      # raises if the HIP library is missing: no fallback
    return nat


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as orc
    return orc


def test_every_statement_kind_on_random_fields(native, oracle):
    _, pd, _ = fas_problem(3, 6)
    hg, ho = native.create(pd), oracle.create(pd)
    rng = np.random.default_rng(11)

    def inner(a):
        b = np.zeros_like(a)
        b[1:-1, 1:-1] = a[1:-1, 1:-1]
        return b

    def load():
        for L in range(3, 7):
            n = (1 << L) + 1
            for f, scale in ((abi.F_SOL, 0.3), (abi.F_RHS, 50.0), (abi.F_RES, 5.0), (abi.F_CORR, 0.3)):
                a = inner(scale * rng.standard_normal((n, n)))
                hg.set_field(L, f, a)
                ho.set_field(L, f, a)

    cases = [("jacobi_newton", abi.make_op(abi.OP_FAS_SMOOTH, 6, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1, 1, 1), 0.8), (6, abi.F_SOL)),
             ("jacobi_newton3", abi.make_op(abi.OP_FAS_SMOOTH, 5, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (3, 1, 1), 0.7), (5, abi.F_SOL)),
             ("jacobi_picard", abi.make_op(abi.OP_FAS_SMOOTH, 6, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (0, 1, 1), 0.6), (6, abi.F_SOL)),
             ("rb_newton2", abi.make_op(abi.OP_FAS_SMOOTH, 6, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (2, 1, 1), 0.9), (6, abi.F_SOL)),
             ("rb_picard", abi.make_op(abi.OP_FAS_SMOOTH, 4, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (0, 1, 1), 1.0), (4, abi.F_SOL)),
             ("residual", abi.make_op(abi.OP_FAS_RESIDUAL, 6, abi.F_RES, abi.F_SOL, abi.F_RHS), (6, abi.F_RES)),
             ("restrict_solution", abi.make_op(abi.OP_FAS_RESTRICT_SOLUTION, 5, abi.F_CORR, abi.F_SOL), (5, abi.F_CORR)),
             ("coarse_rhs", abi.make_op(abi.OP_FAS_COARSE_RHS, 5, abi.F_RHS, abi.F_RES), (5, abi.F_RHS)),
             ("subtract", abi.make_op(abi.OP_FAS_SUBTRACT, 4, abi.F_SOL, abi.F_CORR), (4, abi.F_SOL)),
             ("prolong_add", abi.make_op(abi.OP_PROLONG_ADD, 6, abi.F_SOL, abi.F_SOL, omega=0.95), (6, abi.F_SOL)),
             ("cgs", abi.make_op(abi.OP_FAS_CGS, 3, abi.F_SOL, 0, abi.F_RHS), (3, abi.F_SOL))]
    for name, op, (L, f) in cases:
        load()
        hg.run_ops([op])
        ho.run_ops([op])
        got, want = hg.get_field(L, f), ho.get_field(L, f)
        scale = np.abs(want).max() + 1e-300
        assert np.abs(got - want).max() <= 1e-12 * scale, f"{name}: {np.abs(got - want).max() / scale:.3e}"
        if name == "restrict_solution":
            assert np.array_equal(hg.get_field(L, abi.F_SOL), hg.get_field(L, abi.F_CORR))
    hg.close()
    ho.close()


@pytest.mark.parametrize("name,max_level", [("fas2d_v11_newton", 6), ("fas2d_v21_mixed", 6), ("fas2d_two_grid_picard", 4)])
def test_whole_solves_follow_the_oracle(native, oracle, name, max_level):
    depth, builder = cycles.FAS_CASES[name]
    gen = fas_generator(levels=(max_level - depth, max_level))
    expression, _ = builder(fas_context(gen, depth))
    ops = gen.generate_op_program(expression)
    pd = gen.problem_description()
    hg, ho = native.create(pd), oracle.create(pd)
    got, want = hg.evaluate(ops, 1, 0, 400), ho.evaluate(ops, 1, 0, 400)
    assert got.status == want.status == abi.SOLVE_CONVERGED and got.n_iterations == want.n_iterations
    # 1e-10 relative, down to the rounding floor of the residual evaluation itself: f - A(u) is formed from terms of size
    # |4 / h^2| |u| ~ 1e4, so two correct evaluations (one exp ulp apart) differ by ~1e-12 per node whatever the residual is
    assert np.all(np.abs(got.history - want.history) <= 1e-10 * want.history + 1e-13 * want.history[0])
    a, b = hg.get_field(max_level, abi.F_SOL), ho.get_field(max_level, abi.F_SOL)
    assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max()
    hg.close()
    ho.close()


def test_generate_and_evaluate_at_the_reference_size(native):
    """The drop-in class end to end on the level range of the reference's example (6..10, 1025^2): converges, recovers the
    manufactured solution, and reports (time, convergence factor, iterations) as the reference derives them from stdout."""
    gen = fas_generator(library=native.library(), levels=(6, 10))
    depth = 4
    sm = [[cycles._nw(14)] for _ in range(depth)]
    expression, _ = cycles.fas_v_cycle(fas_context(gen, depth), depth, sm, sm)
    t, c, n = gen.generate_and_evaluate(expression, evaluation_samples=1)
    # coarsest level 65^2, "solved" by 200 Newton-Jacobi sweeps as the template says: slower than with an exact coarse solve
    assert 0 < t < 1e4 and 3 <= n <= 120 and 0.0 < c < 0.85
    sol = gen._handle.get_field(10, abi.F_SOL)
    hh = 1.0 / 1024
    assert np.abs(sol - exact_solution(gen.template, 10)).max() < 0.3 * hh * hh * np.pi ** 2
    gen.close()


@pytest.mark.parametrize("level,sweeps", [(6, 200), (6, 13), (6, 9), (6, 10), (6, 1), (7, 21), (5, 7), (3, 1)])
def test_coarse_sweeps_in_lds_are_bit_identical(native, level, sweeps, monkeypatch):
    """CGS@coarsest: the Newton-Jacobi sweeps on overlapping tiles across 16 / 64 compute units with a rendezvous every nine
    sweeps (evomg_fas_cgs_tiles: 65^2 and 129^2), in one launch with both slots in LDS (evomg_fas_cgs_lds), in one launch
    with the slots in global memory and one launch per sweep: the same formula in the same order, so the same bits —
    sweep counts that are even, odd, one round, one round plus one, a single sweep (the result's slot alternates; the
    slot of the sweep before the last is part of the contract too)."""
    import dataclasses
    _, pd, _ = fas_problem(level, level + 1)
    pd = dataclasses.replace(pd, cgs_max_iterations=sweeps)
    rng = np.random.default_rng(5 + level)
    n = (1 << level) + 1
    u0 = np.zeros((n, n)); u0[1:-1, 1:-1] = 0.3 * rng.standard_normal((n - 2, n - 2))
    f0 = np.zeros((n, n)); f0[1:-1, 1:-1] = 50.0 * rng.standard_normal((n - 2, n - 2))
    op = abi.make_op(abi.OP_FAS_CGS, level, abi.F_SOL, 0, abi.F_RHS)
    # one more Jacobi sweep afterwards reads the slot the coarse sweeps left as "next": its boundary must be intact
    after = [abi.make_op(abi.OP_FAS_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1, 1, 1), 0.8),
             abi.make_op(abi.OP_FAS_RESIDUAL, level, abi.F_RES, abi.F_SOL, abi.F_RHS)]
    results = []
    for knobs in ((), ("EVOMG_NO_FAS_CGS_TILES",), ("EVOMG_NO_FAS_CGS_TILES", "EVOMG_NO_FAS_CGS_LDS"),
                  ("EVOMG_NO_FAS_CGS_TILES", "EVOMG_NO_FAS_CGS_KERNEL")):
        for knob in knobs:
            monkeypatch.setenv(knob, "1")
        h = native.create(pd)
        h.set_field(level, abi.F_SOL, u0)
        h.set_field(level, abi.F_RHS, f0)
        h.run_ops([op])
        first = h.get_field(level, abi.F_SOL)
        h.run_ops(after)
        results.append((first, h.get_field(level, abi.F_SOL), h.get_field(level, abi.F_RES)))
        h.close()
        for knob in knobs:
            monkeypatch.delenv(knob)
    for got in results[1:]:
        assert all(np.array_equal(x, y) for x, y in zip(got, results[0]))
    assert not np.array_equal(results[0][0], u0)


def test_tiled_coarse_sweeps_with_several_candidates_in_flight(native):
    """Six FAS solves at once on one GPU, each with its own handle and stream: the 16-workgroup coarse sweeps of all of them
    meet their rendezvous (or are retried alone), and every solve reports exactly what a solve on its own reports."""
    from concurrent.futures import ThreadPoolExecutor
    gen = fas_generator(library=native.library(), levels=(6, 9))
    depth = 3
    sm = [[cycles._nw(14)] for _ in range(depth)]
    expression, _ = cycles.fas_v_cycle(fas_context(gen, depth), depth, sm, sm)
    ops = gen.generate_op_program(expression)
    pd = gen.problem_description()
    gen.close()
    alone = native.create(pd)
    want = alone.evaluate(ops, 1, 0, 256)
    alone.close()
    handles = [native.create(pd) for _ in range(6)]
    with ThreadPoolExecutor(max_workers=6) as pool:
        outs = list(pool.map(lambda h: [h.evaluate(ops, 1, 0, 256) for _ in range(3)][-1], handles))
    for h in handles:
        h.close()
    assert want.n_iterations >= 3
    for got in outs:
        assert got.status == want.status and got.n_iterations == want.n_iterations
        assert np.array_equal(got.history, want.history)
