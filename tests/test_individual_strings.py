"""stored individuals (`str(individual)`, the replay path of optimization/program.py:904-933 and
scripts/evaluate_evolved_solver.py:40-46) evaluate against our production context into the same DAG — and hence the
same ExaSlang text — as building the candidate programmatically."""
import os

from evostencils_amd.grammar import productions
from helpers import GOLDEN, generator_for, context_for, build_case

# the tutorial's evolved cycle written the way DEAP prints an individual (prefix calls, relaxation-factor indices,
# block shapes as tuples, partitionings by name)
TUTORIAL_INDIVIDUAL = (
    'collective_jacobi_0(28, red_black, residual_0(collective_jacobi_0(31, single, residual_0(update_with'
    '_coarse_grid_correction_0(18, P_1, collective_jacobi_1(13, single, residual_1(update_with_coarse_gri'
    'd_correction_1(33, P_2, update_with_coarse_grid_correction_2(31, P_3, correct_with_coarse_grid_solve'
    'r_3(0, P_4, CGS_4, R_3, residual_3(collective_block_jacobi_3(0, ((1, 6),), residual_3(collective_blo'
    'ck_jacobi_3(5, ((3, 2),), coarsening_2(A_3, zero_3, R_2, coarsening_1(A_2, zero_2, R_1, coarsening_0'
    '(A_1, zero_1, R_0, residual_0(u_and_f))))))))))))))))))')


def test_tutorial_individual_string_compiles_to_the_recorded_cycle():
    gen = generator_for("2D_FD_Poisson_fromL2", 5, 9)
    ctx = context_for(gen, 5, 9)
    expression, rhs = productions.compile_individual(TUTORIAL_INDIVIDUAL, ctx)
    text = gen.generate_cycle_function(expression, None, 5, 9, 9)
    with open(os.path.join(GOLDEN, "tutorial_recorded_cycle.exa3")) as fh:
        assert text.rstrip("\n") == fh.read().rstrip("\n")
    _gen, built, lo, hi = build_case("poisson2d_tutorial")
    assert [o.key() for o in gen.generate_op_program(expression, 5, 9)] == \
           [o.key() for o in _gen.generate_op_program(built, lo, hi)]


def test_individual_strings_are_parsed_not_evaluated():
    """Anything outside the grammar's vocabulary is rejected before it runs (the reference evals these strings)."""
    gen = generator_for("2D_FD_Poisson_fromL2", 6, 8)
    ctx = context_for(gen, 6, 8)
    import pytest
    for text in ("__import__('os').system('true')", "residual_0.__class__", "[c for c in ().__class__.__base__.__subclasses__()]",
                 "residual_0(u_and_f) if 1 else 0", "residual_0(state=u_and_f)", "(lambda: 0)()", "u_and_f[0]", "1.5", "'x'"):
        with pytest.raises(ValueError):
            productions.compile_individual(text, ctx)
    with pytest.raises(ValueError, match="unknown symbol"):
        productions.compile_individual("residual_9(u_and_f)", ctx)
