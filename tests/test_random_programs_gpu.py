"""Randomised parity: (i) random op sequences on random fields, (ii) seeded random candidates built by the grammar's
productions (the C5 generator) — HIP backend vs oracle, bit for bit / iteration for iteration."""
import numpy as np
import pytest

from evostencils_amd import abi, candidates, problems
from helpers import context_for, generator_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()
    return nat


def random_ops(rng, pd, n_ops):
    lo, hi = pd.min_level, pd.max_level
    d = pd.dimension
    shapes = [(2, 1, 1), (1, 2, 1), (2, 2, 1), (3, 2, 1), (1, 3, 1)] if d == 2 else [(2, 1, 1), (1, 1, 2), (2, 2, 1), (2, 2, 2), (1, 3, 2)]
    ops = []
    for _ in range(n_ops):
        kind = rng.integers(0, 8)
        L = int(rng.integers(lo, hi + 1))
        om = float(rng.uniform(0.2, 1.4))
        if kind == 0:
            ops.append(abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1, 1, 1), om))
        elif kind == 1:
            ops.append(abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), om))
        elif kind == 2:
            ops.append(abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, shapes[int(rng.integers(len(shapes)))], om))
        elif kind == 3:
            ops.append(abi.make_op(abi.OP_RESIDUAL, L, abi.F_RES, abi.F_SOL, abi.F_RHS))
            if L > lo and rng.random() < 0.7:      # residual + its restriction: exercises the fused kernel
                ops.append(abi.make_op(abi.OP_RESTRICT, L - 1, abi.F_RHS, abi.F_RES))
        elif kind == 4 and L > lo:
            ops.append(abi.make_op(abi.OP_RESTRICT, L - 1, abi.F_RHS, abi.F_RES))
        elif kind == 5 and L > lo:
            ops.append(abi.make_op(abi.OP_PROLONG_ADD, L, abi.F_SOL, abi.F_SOL, omega=om))
        elif kind == 6:
            ops.append(abi.make_op(abi.OP_SET_ZERO, L, abi.F_RES if rng.random() < 0.5 else abi.F_SOL))
        elif kind == 7 and L > lo:
            ops.append(abi.make_op(abi.OP_PROLONG_SET, L, abi.F_CORR, abi.F_RES))
    return ops


@pytest.mark.parametrize("pname,lo,hi,seed", [("poisson2d", 3, 7, 1), ("poisson2d", 4, 8, 2), ("poisson3d", 2, 5, 3),
                                               ("helmholtz2d", 3, 7, 4)])
def test_random_op_sequences_are_bit_exact(native, pname, lo, hi, seed):
    from oracle import oracle
    from test_parity_gpu import load_random_state
    pd = {"poisson2d": problems.poisson_2d, "poisson3d": problems.poisson_3d,
          "helmholtz2d": lambda a, b: problems.helmholtz_2d(a, b, k=17.0)}[pname](lo, hi)
    hg, ho = native.create(pd), oracle.create(pd)
    rng = np.random.default_rng(seed)
    load_random_state(hg, ho, pd, rng)
    if pd.bc_kind == abi.BC_HELMHOLTZ_ROBIN:
        # make the state boundary-consistent first (the fused RB-GS relies on what every op guarantees afterwards)
        fix = [abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1, 1, 1), 0.5) for L in range(lo, hi + 1)]
        hg.run_ops(fix)
        ho.run_ops(fix)
    ops = random_ops(rng, pd, 60)
    hg.run_ops(ops)
    ho.run_ops(ops)
    for L in range(lo, hi + 1):
        for f in (abi.F_SOL, abi.F_RHS, abi.F_RES):
            assert np.array_equal(hg.get_field(L, f), ho.get_field(L, f)), (L, f)


def test_seeded_candidate_population_matches_oracle(native):
    """A slice of the C5 population at a size the oracle handles: identical iteration counts and convergence factors."""
    from oracle import oracle
    from evostencils_amd.code_generation.hip import ProgramGenerator
    lo, hi = 4, 8
    gen = generator_for("2D_FD_Poisson_fromL2", lo, hi, library=native.library())
    gen.initialize_code_generation(lo, hi)
    pop = candidates.population(lambda: context_for(gen, lo, hi), hi - lo, 2, 12, seed=2024)
    ref = oracle.create(problems.poisson_2d(lo, hi))
    usable = 0
    for expression in pop:
        ops = gen.generate_op_program(expression, lo, hi)
        t, rho, its = gen.generate_and_evaluate(expression, None, lo, hi, "", evaluation_samples=1)
        rt, rrho, rits = ProgramGenerator.interpret_outcome(ref.evaluate(ops, 1, 0, 128), 1e100)
        assert its == rits
        assert rho == pytest.approx(rrho, rel=1e-10)
        usable += its < 1e100
    assert usable >= 6      # the stable-range generator mostly yields convergent solvers


def test_seeded_3d_candidates_match_oracle(native):
    """Seeded 3D candidates on 65^3 with the coarsest grid at 33^3: exercises the fused 3D RB-GS tiles, the cell
    prolongation, static-shape block smoothers and the eight-workgroup coarse CG in whole solves."""
    from oracle import oracle
    from evostencils_amd.code_generation.hip import ProgramGenerator
    lo, hi = 5, 6
    gen = generator_for("3D_FD_Poisson_fromL2", lo, hi, library=native.library())
    gen.initialize_code_generation(lo, hi)
    pop = candidates.population(lambda: context_for(gen, lo, hi), hi - lo, 3, 6, seed=7)
    ref = oracle.create(problems.poisson_3d(lo, hi))
    usable = 0
    for expression in pop:
        ops = gen.generate_op_program(expression, lo, hi)
        t, rho, its = gen.generate_and_evaluate(expression, None, lo, hi, "", evaluation_samples=1)
        rt, rrho, rits = ProgramGenerator.interpret_outcome(ref.evaluate(ops, 1, 0, 128), 1e100)
        assert its == rits
        assert rho == pytest.approx(rrho, rel=1e-10)
        usable += its < 1e100
    assert usable >= 3
