"""State-transition functions of the multigrid grammar, as a plain name -> callable dictionary.

The reference registers these functions as DEAP primitives inside ``add_level``
(evostencils/grammar/multigrid.py:238-385) and DEAP's ``gp.compile`` evaluates an individual's
prefix string against them (optimization/program.py:430). Here they are built without DEAP so that
(i) tests can construct exactly the DAG shapes the grammar can produce, with the same in-place
mutation / shared sub-expression structure, and (ii) stored individuals (``str(individual)``, replay
path optimization/program.py:904-933) can be evaluated into IR DAGs with ``compile_individual``.

Naming of primitives and terminals follows the reference so that its strings evaluate unchanged:
``residual_d``, ``collective_jacobi_d``, ``decoupled_jacobi_d``, ``collective_block_jacobi_d``,
``coarsening_d``, ``update_with_coarse_grid_correction_d``, ``correct_with_coarse_grid_solver_d``,
terminals ``u_and_f``, ``zero_d``, ``A_d``, ``P_d``, ``R_d``, ``CGS_d``, ``single``, ``red_black``; with ``FAS=True``
``jacobi_picard_d`` and ``jacobi_newton_d`` replace the linear smoothers.
"""
import numpy as np

from evostencils_amd.grammar import multigrid as mg
from evostencils_amd.ir import base, system, smoother
from evostencils_amd.ir import partitioning as part


def _level_productions(ctx, depth, operator, coarse_operator, restriction, prolongation, coarse_grid_solver,
                       relaxation_factors, coarsest, FAS=False):
    """Adds the productions of one level (depth 0 = finest) to `ctx`. FAS: the non-linear full-approximation-scheme
    variants (multigrid.py:275-345): the coarse right-hand side gets A_c (R u) added, the coarse result has R u
    subtracted before it is prolongated, and the smoothers are the Picard / Newton point smoothers."""

    def residual(state):
        approximation, rhs = state
        return base.Cycle(approximation, rhs, base.Residual(operator, approximation, rhs),
                          predecessor=approximation.predecessor)

    def _apply(op, cycle):
        cycle.correction = base.Multiplication(op, cycle.correction)
        return cycle

    def _update(index, partitioning_, cycle):
        cycle.relaxation_factor = relaxation_factors[index]
        cycle.partitioning = partitioning_
        return cycle, cycle.rhs

    def _smoothing(index, partitioning_, make_smoother, cycle):
        assert isinstance(cycle.correction, base.Residual), "Invalid production: expected residual"
        _apply(base.Inverse(make_smoother(cycle.correction.operator)), cycle)
        return _update(index, partitioning_, cycle)

    def decoupled_jacobi(index, partitioning_, cycle):
        return _smoothing(index, partitioning_, smoother.generate_decoupled_jacobi, cycle)

    def collective_jacobi(index, partitioning_, cycle):
        return _smoothing(index, partitioning_, smoother.generate_collective_jacobi, cycle)

    def collective_block_jacobi(index, block_shape, cycle):
        return _smoothing(index, part.Single,
                          lambda op: smoother.generate_collective_block_jacobi(op, block_shape), cycle)

    def _restrict(restriction_operator, cycle):
        if not FAS:
            return _apply(restriction_operator, cycle)
        residual_c = base.mul(restriction_operator, cycle.correction)
        residual_fas = base.mul(coarse_operator, base.Multiplication(restriction_operator, cycle.approximation))
        cycle.correction = base.add(residual_c, residual_fas)
        return cycle

    def jacobi_picard(index, partitioning_, cycle):
        return _smoothing(index, partitioning_, smoother.generate_jacobi_picard, cycle)

    def jacobi_newton(index, partitioning_, n_newton_steps, cycle):
        return _smoothing(index, partitioning_, lambda op: smoother.generate_jacobi_newton(op, n_newton_steps), cycle)

    def coarsening(coarse_op, coarse_approximation, restriction_operator, cycle):
        _restrict(restriction_operator, cycle)
        new_cycle = base.Cycle(coarse_approximation, cycle.correction,
                               base.Residual(coarse_op, coarse_approximation, cycle.correction))
        new_cycle.predecessor = cycle
        return new_cycle

    def update_with_coarse_grid_correction(index, prolongation_operator, state, restriction_operator=None):
        coarse_cycle = state[0]
        fine_cycle = coarse_cycle.predecessor
        if FAS:
            restricted = base.mul(restriction_operator, fine_cycle.approximation)        # subtracted for FAS
            fine_cycle.correction = base.mul(prolongation_operator, base.sub(coarse_cycle, restricted))
        else:
            fine_cycle.correction = base.Multiplication(prolongation_operator, coarse_cycle)
        return _update(index, part.Single, fine_cycle)

    def correct_with_coarse_grid_solver(index, prolongation_operator, cgs, restriction_operator, cycle):
        _restrict(restriction_operator, cycle)
        if FAS:
            approximation_c = base.mul(cgs, cycle.correction)
            restricted = base.mul(restriction_operator, cycle.approximation)
            cycle.correction = base.mul(prolongation_operator, base.sub(approximation_c, restricted))
        else:
            _apply(cgs, cycle)
            _apply(prolongation_operator, cycle)
        return _update(index, part.Single, cycle)

    ctx[f"residual_{depth}"] = residual
    if FAS:
        ctx[f"jacobi_picard_{depth}"] = jacobi_picard
        ctx[f"jacobi_newton_{depth}"] = jacobi_newton
    else:
        ctx[f"decoupled_jacobi_{depth}"] = decoupled_jacobi
        ctx[f"collective_jacobi_{depth}"] = collective_jacobi
        ctx[f"collective_block_jacobi_{depth}"] = collective_block_jacobi
    ctx[prolongation.name] = prolongation
    ctx[restriction.name] = restriction
    if not coarsest:
        ctx[f"zero_{depth + 1}"] = system.ZeroApproximation(coarse_operator.grid)
        ctx[f"A_{depth + 1}"] = coarse_operator
        ctx[f"update_with_coarse_grid_correction_{depth}"] = update_with_coarse_grid_correction
        ctx[f"coarsening_{depth}"] = coarsening
    else:
        ctx[f"correct_with_coarse_grid_solver_{depth}"] = correct_with_coarse_grid_solver
        ctx[f"CGS_{depth + 1}"] = coarse_grid_solver


def generate_context(approximation, rhs, dimension, coarsening_factors, max_level, equations, operators, fields,
                     relaxation_factor_samples=37, depth=2, FAS=False):
    """Evaluation context for a `depth`+1-level hierarchy rooted at `max_level` (the counterpart of
    generate_primitive_set, multigrid.py:409-478). Level `max_level - depth` is the coarsest one, solved by
    the coarse-grid solver."""
    assert depth >= 1, "The maximum number of levels must be greater zero"
    relaxation_factors = np.linspace(0.1, 1.9, relaxation_factor_samples)     # multigrid.py:428
    ctx = {"u_and_f": (approximation, rhs), "single": part.Single, "red_black": part.RedBlack,
           "relaxation_factors": relaxation_factors}
    fine_grid = approximation.grid
    for d in range(depth):
        level = max_level - d
        coarse_grid = system.get_coarse_grid(fine_grid, coarsening_factors)
        operator, restriction, prolongation = mg.generate_operators_from_l2_information(
            equations, operators, fields, level, d, fine_grid, coarse_grid)
        coarse_operator = mg.generate_system_operator_from_l2_information(
            equations, operators, fields, level - 1, d + 1, coarse_grid)
        cgs = base.CoarseGridSolver("Coarse-Grid Solver", coarse_operator, None)
        _level_productions(ctx, d, operator, coarse_operator, restriction, prolongation, cgs,
                           relaxation_factors, coarsest=(d == depth - 1), FAS=FAS)
        fine_grid = coarse_grid
    return ctx


def compile_individual(individual_string: str, context: dict):
    """DAG of a stored individual, e.g. `str(individual)` as written to hof/individual_<j>.txt
    (scripts/optimize.py:176-179). Returns the (Cycle expression, rhs) state like gp.compile does.

    The string is parsed, not evaluated: only calls of names found in `context`, those names themselves, integer
    constants and tuples of them are accepted (the whole vocabulary DEAP prints for this grammar). The reference
    hands such strings to `eval` (gp.compile; optimization/program.py:904-933), which would execute whatever a
    tampered checkpoint or hall-of-fame file contains."""
    import ast
    env = {k: v for k, v in context.items() if k != "relaxation_factors"}

    def value(node):
        if isinstance(node, ast.Call):
            if not isinstance(node.func, ast.Name) or node.keywords:
                raise ValueError("individual string: only plain calls of grammar productions are allowed")
            fn = value(node.func)
            if not callable(fn):
                raise ValueError(f"individual string: {node.func.id!r} is not a production")
            return fn(*[value(a) for a in node.args])
        if isinstance(node, ast.Name):
            if node.id not in env:
                raise ValueError(f"individual string: unknown symbol {node.id!r}")
            return env[node.id]
        if isinstance(node, ast.Constant) and type(node.value) is int:
            return node.value
        if isinstance(node, ast.Tuple):
            return tuple(value(e) for e in node.elts)
        raise ValueError(f"individual string: {type(node).__name__} is not part of the grammar's vocabulary")

    tree = ast.parse(individual_string.strip(), mode="eval")
    return value(tree.body)
