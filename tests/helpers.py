"""Shared test plumbing: contexts, generators and op programs for the cases in tests/cycles.py."""
import os

from evostencils_amd import abi, candidates, problems
from evostencils_amd.code_generation.hip import ProgramGenerator
from evostencils_amd.grammar import productions
from evostencils_amd.ir import base, system

import cycles

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _NoLibrary:
    """Stands in for the native library where a test only lowers (no GPU needed)."""


def generator_for(problem_name, min_level, max_level, library=None, **kw):
    """Generators of the test-suite run `with jacobi` sweeps with their textbook meaning unless a test asks for the
    reference pipeline's ("exastencils-v1.1", the ProgramGenerator default): otherwise every Jacobi / block-Jacobi
    kernel check would compare two untouched fields."""
    kw.setdefault("jacobi_semantics", "textbook")
    pd = problems.BY_NAME[problem_name](min_level, max_level)
    return ProgramGenerator(problem=pd, library=library or _NoLibrary(), **kw)


def context_for(gen: ProgramGenerator, min_level, max_level):
    return candidates.production_context(gen, min_level, max_level)


def window_context(gen: ProgramGenerator, min_level, max_level):
    """Production context of a level window below the problem's finest level, as the optimizer builds it for the later
    runs of a multi-run optimisation (optimization/program.py:786-789,811-830: coarse approximations / right-hand sides,
    `generate_primitive_set(..., max_level, depth=levels_per_run)`)."""
    approximation = system.Approximation("x", [base.Approximation(f.name, g) for f, g in zip(gen.fields, gen.finest_grid)])
    top = [eq for eq in gen.equations if eq.level == gen.max_level]
    rhs = system.RightHandSide("b", [base.RightHandSide(eq.rhs_name, g) for eq, g in zip(top, gen.finest_grid)])
    for _ in range(gen.max_level - max_level):
        approximation = system.get_coarse_approximation(approximation, gen.coarsening_factor)
        rhs = system.get_coarse_rhs(rhs, gen.coarsening_factor)
    return productions.generate_context(approximation, rhs, gen.dimension, gen.coarsening_factor, max_level,
                                        gen.equations, gen.operators, gen.fields, depth=max_level - min_level)


def build_case(name, library=None, **kw):
    """(generator, expression, min_level, max_level) for a named case of cycles.CASES."""
    problem, min_level, max_level, builder = cycles.CASES[name]
    gen = generator_for(problem, min_level, max_level, library, **kw)
    ctx = context_for(gen, min_level, max_level)
    expression, _ = builder(ctx)
    return gen, expression, min_level, max_level


def ops_for_case(name):
    gen, expression, lo, hi = build_case(name)
    return gen, gen.generate_op_program(expression, lo, hi)
