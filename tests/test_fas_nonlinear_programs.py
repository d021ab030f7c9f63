"""Non-linear FAS problems whose stencil entry is NOT the reference example's gam * exp(Solution) (round 6, ABI 6): the
reference's parser admits any sympy term (code_generation/parser.py:146-169) and its FAS generator differentiates it with
sympy (exastencils_FAS.py); here the entry and its Newton term travel as two reverse-Polish programs
(include/evomg.h: evomg_nl_program_t; compiled by code_generation/fas.py). CPU half: the compiler against sympy, the
descriptor, the oracle on two problems of this repository (tests/golden/problems/FAS_2D_Cubic, FAS_2D_Saturating: known
answers), and the reference example run THROUGH the program path against its dedicated path."""
import dataclasses
import math

import numpy as np
import pytest
import sympy

from evostencils_amd import abi
from evostencils_amd.code_generation import fas
from oracle import oracle

from test_fas_oracle import exact_solution, fas_problem, ops_for

ENTRIES = ["( gam * exp( Solution<active>@current ) )",
           "( gam * ( Solution<active>@current ) * ( Solution<active>@current ) )",
           "( gam / ( 1.0 + ( Solution<active>@current ) * ( Solution<active>@current ) ) )",
           "( gam * ( 1.0 + sin( Solution<active>@current ) ) )",
           "( gam * tanh( Solution<active>@current ) * ( Solution<active>@current ) )",
           "( gam * sqrt( 1.0 + ( Solution@current ) ^ 2 ) )",
           "( 2.5 * cos( Solution@current ) - gam * log( 2.0 + ( Solution@current ) ^ 2 ) )"]


@pytest.mark.parametrize("text", ENTRIES)
def test_programs_evaluate_what_sympy_says(text):
    S = fas.SOLUTION_SYMBOL
    sigma_expr = fas.nonlinear_expression(text)
    sigma, jacobian, consts = fas.compile_nonlinear(sigma_expr, {"gam": 7.5})
    jac_expr = sympy.simplify(sympy.diff(sigma_expr * S, S))          # exastencils_FAS.py: sympy_jacobian
    assert max(len(sigma), len(jacobian)) <= abi.NL_MAX_OPS and len(consts) <= abi.NL_MAX_CONSTS
    assert max(fas._stack_depth(sigma), fas._stack_depth(jacobian)) <= abi.NL_MAX_STACK
    for u in (-1.3, -0.2, 0.0, 0.4, 1.7):
        want_s = float(sigma_expr.subs({S: u, "gam": 7.5}))
        want_j = float(jac_expr.subs({S: u, "gam": 7.5}))
        assert abs(fas.evaluate_program(sigma, consts, u) - want_s) <= 1e-13 * max(1.0, abs(want_s))
        assert abs(fas.evaluate_program(jacobian, consts, u) - want_j) <= 1e-12 * max(1.0, abs(want_j))
        # ... and the Newton term IS the derivative of N(u) = sigma(u) u
        d = 1e-6
        n = lambda v: float(sigma_expr.subs({S: v, "gam": 7.5})) * v
        assert abs((n(u + d) - n(u - d)) / (2 * d) - want_j) <= 1e-5 * max(1.0, abs(want_j))


def test_the_reference_example_keeps_its_dedicated_kernels_and_everything_else_becomes_a_program():
    assert fas.classify_nonlinear(ENTRIES[0], 10.0) == (abi.NONLINEAR_GAM_U_EXP_U, None)
    assert fas.classify_nonlinear("( exp( Solution@current ) * gam )", 10.0)[0] == abi.NONLINEAR_GAM_U_EXP_U
    for text in ENTRIES[1:]:
        kind, program = fas.classify_nonlinear(text, 10.0)
        assert kind == abi.NONLINEAR_PROGRAM and program is not None
    with pytest.raises(fas.UnsupportedFASExpression):
        fas.classify_nonlinear("( gam * atan( Solution@current ) )", 1.0)
    with pytest.raises(fas.UnsupportedFASExpression):
        fas.classify_nonlinear("( gam * ( Solution@current ) ^ 0.3 )", 1.0)
    with pytest.raises(fas.UnsupportedFASExpression):
        fas.classify_nonlinear("( kappa * Solution@current )", 1.0)      # a symbol the template does not define
    with pytest.raises(fas.UnsupportedFASExpression):
        fas.classify_nonlinear("( beta * Solution@current )", 1.0)       # ... or one sympy reads as something else


def test_descriptor_carries_the_programs():
    _, pd, t = fas_problem(3, 5, problem="FAS_2D_Cubic")
    assert t.gamma == 40.0 and pd.nonlinear_kind == abi.NONLINEAR_PROGRAM
    c = pd.to_c()
    q = c.nonlinear_program
    assert c.abi_version == 6 and c.nonlinear_kind == abi.NONLINEAR_PROGRAM
    sigma, jacobian, consts = pd.nonlinear_program
    assert list(q.sigma[:q.n_sigma]) == sigma and list(q.jacobian[:q.n_jacobian]) == jacobian
    assert list(q.consts[:len(consts)]) == consts
    # N(u) = gam u^3: sigma = gam u u, Newton term 3 gam u^2
    assert abs(fas.evaluate_program(sigma, consts, 0.5) - 40.0 * 0.25) < 1e-14
    assert abs(fas.evaluate_program(jacobian, consts, 0.5) - 3 * 40.0 * 0.25) < 1e-13
    # a malformed program is refused by the library's descriptor check, not executed
    bad = dataclasses.replace(pd, nonlinear_program=([abi.NL_ADD], jacobian, consts))
    with pytest.raises(abi.BackendError):
        oracle.create(bad)


@pytest.mark.parametrize("problem", ["FAS_2D_Cubic", "FAS_2D_Saturating"])
@pytest.mark.parametrize("max_level", [5, 6])
def test_oracle_recovers_the_manufactured_solution(problem, max_level):
    _, pd, t = fas_problem(3, max_level, problem=problem)
    h = oracle.create(pd)
    out = h.evaluate(ops_for(3, max_level, problem=problem), 1, 0, 400)
    assert out.status == abi.SOLVE_CONVERGED and 3 <= out.n_iterations <= 40
    err = np.abs(h.get_field(max_level, abi.F_SOL) - exact_solution(t, max_level)).max()
    hh = 1.0 / (1 << max_level)
    assert err < 0.3 * hh * hh * (np.pi ** 2), err
    rates = out.history[1:] / out.history[:-1]
    assert rates[2:].max() < 0.6
    h.close()


def test_the_example_through_the_program_path_is_the_example():
    """gam * exp(Solution) compiled to programs and run by the interpreter against the dedicated formulas: the same solve
    (the Newton term is evaluated in sympy's order instead of the template's: agreement to rounding, not bit for bit)."""
    _, pd, _ = fas_problem(3, 6)
    program = fas.compile_nonlinear(fas.nonlinear_expression(ENTRIES[0]), {"gam": pd.nonlinear_gamma})
    via_program = dataclasses.replace(pd, nonlinear_kind=abi.NONLINEAR_PROGRAM, nonlinear_program=program)
    ops = ops_for(3, 6)
    a, b = oracle.create(pd), oracle.create(via_program)
    ra, rb = a.evaluate(ops, 1, 0, 400), b.evaluate(ops, 1, 0, 400)
    assert ra.status == rb.status == abi.SOLVE_CONVERGED and ra.n_iterations == rb.n_iterations
    assert np.all(np.abs(ra.history - rb.history) <= 1e-10 * ra.history + 1e-13 * ra.history[0])
    assert np.abs(a.get_field(6, abi.F_SOL) - b.get_field(6, abi.F_SOL)).max() < 1e-12
    a.close()
    b.close()
