"""GPU side of the pin: the HIP backend, driven through the drop-in ProgramGenerator with its DEFAULT settings,
returns for the reference's recorded cycle exactly the triple the reference recorded
(notebooks/tutorial.ipynb:3373, tests/golden/tutorial_recorded_result.json), and follows the oracle in the
reference pipeline's `with jacobi` semantics (EVOMG_JACOBI_EXASTENCILS_V11) op by op and solve by solve."""
import json
import os

import numpy as np
import pytest

from evostencils_amd import abi, candidates, problems
from evostencils_amd.code_generation.hip import ProgramGenerator
import cycles
from helpers import GOLDEN, build_case, context_for

pytestmark = pytest.mark.gpu

RECORDED = json.load(open(os.path.join(GOLDEN, "tutorial_recorded_result.json")))


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()
    return nat


def test_generate_and_evaluate_returns_the_recorded_triple(native):
    pd = problems.poisson_2d(RECORDED["min_level"], RECORDED["max_level"])
    gen = ProgramGenerator(problem=pd, library=native.library(), solver_iteration_limit=RECORDED["solver_iteration_limit"])
    assert gen.jacobi_semantics == "exastencils-v1.1"            # the default IS the reference pipeline's behaviour
    lo, hi = pd.min_level, pd.max_level
    gen.initialize_code_generation(lo, hi)
    expression, _ = cycles.tutorial_cycle(context_for(gen, lo, hi))
    t, rho, its = gen.generate_and_evaluate(expression, gen.generate_storage(lo, hi, gen.finest_grid), lo, hi, "",
                                            evaluation_samples=RECORDED["evaluation_samples"])
    assert its == RECORDED["number_of_iterations"]
    assert rho == RECORDED["convergence_factor"]                 # to the last digit
    assert 0 < t < RECORDED["solving_time_ms"]                   # 100 iterations at 513^2: faster than the recorded CPU run
    gen.close()


@pytest.mark.parametrize("name", ["poisson2d_tutorial", "poisson2d_L7_blocks", "poisson2d_L8_v11_mixed", "poisson3d_L4_mixed"])
def test_reference_semantics_follow_the_oracle(native, name):
    from oracle import oracle
    gen, expression, lo, hi = build_case(name, library=native.library(), jacobi_semantics="exastencils-v1.1")
    gen.initialize_code_generation(lo, hi)
    ops = gen.generate_op_program(expression, lo, hi)
    ho = oracle.create(gen.problem)
    want = ho.evaluate(ops, 1, 0, 128)
    got = gen._handle.evaluate(ops, 1, 0, 128)
    assert got.status == want.status and got.n_iterations == want.n_iterations
    floor = want.history > 1e-9 * want.history[0]
    assert np.max(np.abs(got.history - want.history)[floor] / want.history[floor]) <= 1e-10
    assert np.array_equal(gen._handle.get_field(hi, abi.F_SOL), ho.get_field(hi, abi.F_SOL))
    gen.close()


@pytest.mark.parametrize("dim,level", [(2, 6), (2, 4), (3, 4), (3, 7)])
def test_lost_sweeps_leave_the_field_alone(native, dim, level):
    """Every kernel family behind a Single-partitioned SMOOTH (strip, tile, generic, block fast/generic) runs and is
    discarded; the red-black sweep of the same level takes effect."""
    pd = (problems.poisson_2d if dim == 2 else problems.poisson_3d)(level - 1, level)
    h = native.create(pd, jacobi_semantics="exastencils-v1.1")
    rng = np.random.default_rng(level)
    n = (1 << level) + 1
    x = rng.standard_normal((n,) * dim)
    f = rng.standard_normal((n,) * dim)
    blocks = [(1, 1, 1), (2, 1, 1), (1, 2, 1), (2, 2, 1), (3, 2, 1)] + ([(2, 2, 2), (1, 1, 3)] if dim == 3 else [])
    for block in blocks:
        h.set_field(level, abi.F_SOL, x)
        h.set_field(level, abi.F_RHS, f)
        # twice: the second sweep meets the spare array the first one left behind
        h.run_ops([abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, block, 0.8)] * 2)
        assert np.array_equal(h.get_field(level, abi.F_SOL), x), block
    h.run_ops([abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 0.8)])
    assert not np.array_equal(h.get_field(level, abi.F_SOL), x)
    h.close()


def test_seeded_candidates_in_reference_semantics(native):
    """A slice of the seeded population (candidates.py) through generate_and_evaluate with default settings vs the
    oracle in the same semantics: identical iteration counts and reported factors."""
    from oracle import oracle
    lo, hi = 5, 8
    gen = ProgramGenerator(problem=problems.poisson_2d(lo, hi), library=native.library(), solver_iteration_limit=500)
    gen.initialize_code_generation(lo, hi)
    ho = oracle.create(gen.problem)
    for expression in candidates.population(lambda: context_for(gen, lo, hi), hi - lo, 2, 8, seed=77):
        t, rho, its = gen.generate_and_evaluate(expression, None, lo, hi, "", evaluation_samples=1)
        ops = gen.generate_op_program(expression, lo, hi)
        _, want_rho, want_its = ProgramGenerator.interpret_outcome(ho.evaluate(ops, 1, 0, 128), 1e100, 500)
        assert its == want_its
        assert rho == pytest.approx(want_rho, rel=1e-9)
    gen.close()
