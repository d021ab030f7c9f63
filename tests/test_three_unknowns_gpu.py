"""Systems of THREE PDEs on the GPU (round 6, ABI 6): the three-species problem through the C-ABI against the CPU oracle —
every op kind bit for bit on random fields (collective / decoupled / block smoothers over one, two and three of the
unknowns), the stacked coarse-grid CG, whole solves of the golden-string candidates in both `with jacobi` semantics through
the drop-in ProgramGenerator, the file-defined problem, and the launch fusions switched off."""
import os

import numpy as np
import pytest

from evostencils_amd import abi, problems
from evostencils_amd.code_generation.hip import ProgramGenerator

import cycles
from helpers import build_case, context_for, ops_for_case
from test_three_unknowns import FIXTURES, NF, direct_solution, v_cycle_ops

pytestmark = pytest.mark.gpu
F = abi.field_of


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()
    return nat


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as orc
    return orc


def load_random(handles, level, seed):
    rng = np.random.default_rng(seed)
    n = (1 << level) + 1
    for slot in (abi.F_SOL, abi.F_RHS, abi.F_RES):
        for u in range(NF):
            a = rng.standard_normal((n, n)) * (1e4 if slot == abi.F_RHS else 1.0)
            for h in handles:
                h.set_field(level, F(slot, u), a)


@pytest.mark.parametrize("level", [4, 6])
def test_every_op_kind_is_bit_identical(native, oracle, level):
    pd = problems.three_species_2d(level - 1, level)
    hg, ho = native.create(pd), oracle.create(pd)
    smooths = [(0b111, (1, 1, 1), abi.PART_SINGLE), (0b111, (1, 1, 1), abi.PART_RED_BLACK), (0b001, (1, 1, 1), abi.PART_SINGLE),
               (0b010, (1, 1, 1), abi.PART_RED_BLACK), (0b100, (1, 1, 1), abi.PART_SINGLE), (0b101, (1, 1, 1), abi.PART_SINGLE),
               (0b110, (1, 1, 1), abi.PART_RED_BLACK), (0b111, (2, 1, 1), abi.PART_SINGLE), (0b011, (2, 2, 1), abi.PART_SINGLE),
               (0b111, (1, 2, 1), abi.PART_SINGLE), (0b100, (3, 2, 1), abi.PART_SINGLE)]
    for k, (mask, block, part) in enumerate(smooths):
        load_random((hg, ho), level, 100 * level + k)
        first = [u for u in range(NF) if (mask >> u) & 1][0]
        ops = [abi.make_op(abi.OP_SMOOTH, level, F(abi.F_SOL, first), 0, F(abi.F_RHS, first), part, block, 0.85, unknowns=mask)] * 2
        ops += [abi.make_op(abi.OP_RESIDUAL, level, F(abi.F_RES, u), F(abi.F_SOL), F(abi.F_RHS, u)) for u in range(NF)]
        ops += [abi.make_op(abi.OP_RESTRICT, level - 1, F(abi.F_RHS, u), F(abi.F_RES, u)) for u in range(NF)]
        ops += [abi.make_op(abi.OP_SET_ZERO, level - 1, F(abi.F_SOL, u)) for u in range(NF)]
        ops += [abi.make_op(abi.OP_SMOOTH, level - 1, F(abi.F_SOL), 0, F(abi.F_RHS), abi.PART_RED_BLACK, (1, 1, 1), 1.1, unknowns=0b111)]
        ops += [abi.make_op(abi.OP_PROLONG_ADD, level, F(abi.F_SOL, u), F(abi.F_SOL, u), omega=0.9) for u in range(NF)]
        ops += [abi.make_op(abi.OP_PROLONG_SET, level, F(abi.F_CORR, u), F(abi.F_SOL, u)) for u in range(NF)]
        hg.run_ops(ops)
        ho.run_ops(ops)
        for L, slots in ((level, (abi.F_SOL, abi.F_RES, abi.F_CORR)), (level - 1, (abi.F_SOL, abi.F_RHS))):
            for slot in slots:
                for u in range(NF):
                    assert np.array_equal(hg.get_field(L, F(slot, u)), ho.get_field(L, F(slot, u))), (mask, block, part, L, slot, u)
    # three unknowns with a block of (2, 2): twelve cells, more than a local system may have — rejected on both sides
    too_big = abi.make_op(abi.OP_SMOOTH, level, F(abi.F_SOL), 0, F(abi.F_RHS), abi.PART_SINGLE, (2, 2, 1), 0.8, unknowns=0b111)
    assert hg.try_evaluate([too_big], 1, 0, 8)[0] != abi.OK and ho.try_evaluate([too_big], 1, 0, 8)[0] != abi.OK
    hg.close()
    ho.close()


@pytest.mark.parametrize("lo", [3, 4, 5])
def test_stacked_coarse_cg(native, oracle, lo):
    pd = problems.three_species_2d(lo, lo + 1)
    hg, ho = native.create(pd), oracle.create(pd)
    rng = np.random.default_rng(lo)
    n = (1 << lo) + 1
    ops = []
    for u in range(NF):
        b = rng.standard_normal((n, n))
        for h in (hg, ho):
            h.set_field(lo, F(abi.F_CGS_RHS, u), b)
        ops.append(abi.make_op(abi.OP_SET_ZERO, lo, F(abi.F_CORR, u)))
    ops.append(abi.make_op(abi.OP_COARSE_SOLVE, lo))
    hg.run_ops(ops)
    ho.run_ops(ops)
    for u in range(NF):
        assert np.array_equal(hg.get_field(lo, F(abi.F_CORR, u)), ho.get_field(lo, F(abi.F_CORR, u)))
    res = [abi.make_op(abi.OP_RESIDUAL, lo, F(abi.F_RES, u), F(abi.F_CORR), F(abi.F_CGS_RHS, u)) for u in range(NF)]
    hg.run_ops(res)
    r = np.sqrt(sum(hg.norm(lo, F(abi.F_RES, u)) ** 2 for u in range(NF)))
    b = np.sqrt(sum(hg.norm(lo, F(abi.F_CGS_RHS, u)) ** 2 for u in range(NF)))
    assert r <= 50 * pd.cgs_target_res_reduction * b
    hg.close()
    ho.close()


@pytest.mark.parametrize("semantics", ["textbook", "exastencils-v1.1"])
@pytest.mark.parametrize("name", ["threespecies2d_L5_mixed", "threespecies2d_L6_v21_rb"])
def test_whole_solves_through_the_program_generator(native, oracle, name, semantics):
    gen, expression, lo, hi = build_case(name, library=native.library(), jacobi_semantics=semantics)
    gen.initialize_code_generation(lo, hi)
    t, rho, its = gen.generate_and_evaluate(expression, gen.generate_storage(lo, hi, gen.finest_grid), lo, hi, "", evaluation_samples=1)
    ops = gen.generate_op_program(expression, lo, hi)
    ho = oracle.create(gen.problem)
    want = ho.evaluate(ops, 1, 0, 128)
    got = gen._handle.evaluate(ops, 1, 0, 128)
    _, want_rho, want_its = ProgramGenerator.interpret_outcome(want, 1e100)
    assert its == want_its and rho == pytest.approx(want_rho, rel=1e-9) and t > 0
    assert got.status == want.status and got.n_iterations == want.n_iterations
    # the loop's norm is summed in the device's tree order here and in plain row order there: 1e-10, as for two unknowns;
    # the fields themselves — every kernel of this path and the stacked CG are bit-identical to the oracle's loops — are equal
    floor = want.history > 1e-9 * want.history[0]
    assert np.max(np.abs(got.history - want.history)[floor] / want.history[floor]) <= 1e-10
    for u in range(NF):
        assert np.array_equal(gen._handle.get_field(hi, F(abi.F_SOL, u)), ho.get_field(hi, F(abi.F_SOL, u)))
    ho.close()
    gen.close()


def test_the_file_defined_problem_at_257_squared(native):
    """The problem as its files define it (reference-style constructor), levels 4..8, the problem's own solver settings as the
    candidate: converges, and the solution is the sparse direct solve's."""
    stem = "ThreeSpecies/2D_FD_ThreeSpecies"
    gen = ProgramGenerator("unused.jar", FIXTURES, stem + ".settings", stem + ".knowledge", "unused.platform", jacobi_semantics="textbook")
    lo, hi = 4, 7
    gen.reinitialize(lo, hi, {})
    ctx = context_for(gen, lo, hi)
    rb = [("rbgs", 20)]
    expression, _ = cycles.v_cycle(ctx, hi - lo, [rb * 2] * (hi - lo), [rb] * (hi - lo), cgc_index=18)
    gen.initialize_code_generation(lo, hi)
    t, rho, its = gen.generate_and_evaluate(expression, gen.generate_storage(lo, hi, gen.finest_grid), lo, hi, "", evaluation_samples=1)
    assert 0 < t < 1e4 and rho < 0.2 and its <= 20
    want = direct_solution(gen.problem, hi)
    for u in range(NF):
        got = gen._handle.get_field(hi, F(abi.F_SOL, u))
        assert np.abs(got - want[u]).max() <= 1e-8 * max(np.abs(want[u]).max(), 1e-300)
    gen.close()


def test_one_launch_per_statement_for_all_three_unknowns_does_not_change_a_bit(native, monkeypatch):
    gen, ops = ops_for_case("threespecies2d_L5_mixed")
    _, lo, hi, _ = cycles.CASES["threespecies2d_L5_mixed"]
    pd = problems.three_species_2d(lo, hi)
    results = []
    for plain in (False, True):
        for k in ("EVOMG_NO_SYS_CG_SMALL", "EVOMG_NO_SYS_PAIR_FUSE"):
            if plain:
                monkeypatch.setenv(k, "1")
            else:
                monkeypatch.delenv(k, raising=False)
        h = native.create(pd)
        out = h.evaluate(ops, 1, 0, 128)
        results.append((out, [h.get_field(hi, F(abi.F_SOL, u)) for u in range(NF)]))
        h.close()
    (a, xa), (b, xb) = results
    assert a.status == b.status and a.n_iterations == b.n_iterations and a.n_iterations > 3
    assert np.array_equal(a.history, b.history)
    for p, q in zip(xa, xb):
        assert np.array_equal(p, q)


@pytest.mark.parametrize("shapes", [((2, 1, 1), (1, 1, 1), (1, 2, 1)), ((1, 1, 1), (2, 2, 1), (2, 1, 1)), ((3, 1, 1), (1, 3, 1), (1, 1, 1))])
def test_three_different_block_shapes_run_only_where_their_result_is_never_used(native, oracle, shapes):
    """evomg_op_t.block / block_other / block_third (ABI 6) with differing shapes — what most BlockShape terminals of a
    three-unknown grammar look like (grammar/multigrid.py:388-407): ERR_UNSUPPORTED under textbook semantics, launched for its
    time under the reference pipeline's, where every field stays bit for bit what it was (as for two unknowns)."""
    level = 5
    pd = problems.three_species_2d(level - 1, level)
    op = abi.make_op(abi.OP_SMOOTH, level, F(abi.F_SOL), 0, F(abi.F_RHS), abi.PART_SINGLE, shapes[0], 0.85, unknowns=0b111,
                     block_other=shapes[1], block_third=shapes[2])
    for sem in ("textbook", "exastencils-v1.1"):
        hg, ho = native.create(pd, jacobi_semantics=sem), oracle.create(pd, jacobi_semantics=sem)
        load_random((hg, ho), level, 78)
        before = [hg.get_field(level, F(abi.F_SOL, u)) for u in range(NF)]
        if sem == "textbook":
            assert hg.try_evaluate([op], 1, 0, 8)[0] == ho.try_evaluate([op], 1, 0, 8)[0] == abi.ERR_UNSUPPORTED
        else:
            hg.run_ops([op, op])
            ho.run_ops([op, op])
            for u in range(NF):
                assert np.array_equal(hg.get_field(level, F(abi.F_SOL, u)), before[u])
                assert np.array_equal(ho.get_field(level, F(abi.F_SOL, u)), before[u])
        hg.close()
        ho.close()
    # more than eight cells in all: refused on both sides whatever the semantics
    big = abi.make_op(abi.OP_SMOOTH, level, F(abi.F_SOL), 0, F(abi.F_RHS), abi.PART_SINGLE, (2, 2, 1), 0.85, unknowns=0b111,
                      block_other=(2, 1, 1), block_third=(3, 1, 1))
    hg, ho = native.create(pd, jacobi_semantics="exastencils-v1.1"), oracle.create(pd, jacobi_semantics="exastencils-v1.1")
    assert hg.try_evaluate([big], 1, 0, 8)[0] != abi.OK and ho.try_evaluate([big], 1, 0, 8)[0] != abi.OK
    hg.close()
    ho.close()
