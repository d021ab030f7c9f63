"""Golden-string tests of the lowering: our IR + our walker + our ExaSlang renderer must print,
character for character, what the reference's ProgramGenerator.generate_cycle_function prints for
the same candidate (vectors captured from the reference by tests/golden/generate_golden.py), and
what the reference's own tutorial notebook recorded (notebooks/tutorial.ipynb:3421-3470)."""
import json
import os

import pytest

import cycles
from helpers import GOLDEN, build_case

CASE_NAMES = sorted(cycles.CASES)


@pytest.mark.parametrize("name", CASE_NAMES)
def test_cycle_function_matches_reference(name):
    gen, expression, lo, hi = build_case(name)
    storages = gen.generate_storage(lo, hi, gen.finest_grid)
    text = gen.generate_cycle_function(expression, storages, lo, hi, hi)
    with open(os.path.join(GOLDEN, f"{name}.exa3")) as fh:
        expected = fh.read()
    assert text == expected


def test_tutorial_cycle_matches_recorded_notebook_output():
    gen, expression, lo, hi = build_case("poisson2d_tutorial")
    text = gen.generate_cycle_function(expression, None, lo, hi, hi)
    with open(os.path.join(GOLDEN, "tutorial_recorded_cycle.exa3")) as fh:
        recorded = fh.read()
    assert text.rstrip("\n") == recorded.rstrip("\n")


@pytest.mark.parametrize("name", CASE_NAMES)
def test_lowering_is_repeatable_and_restores_valid_flags(name):
    """exastencils.py:330-334: a second lowering of the same (shared, mutated) DAG gives the same text."""
    gen, expression, lo, hi = build_case(name)
    a = gen.generate_cycle_function(expression, None, lo, hi, hi)
    b = gen.generate_cycle_function(expression, None, lo, hi, hi)
    assert a == b


@pytest.mark.parametrize("name", CASE_NAMES)
def test_one_op_per_statement(name):
    """The op program is the statement list, 1:1 and in order (self-copies are kept as COPY ops)."""
    gen, expression, lo, hi = build_case(name)
    stmts = gen.lower(expression, lo, hi)
    ops = gen.generate_op_program(expression, lo, hi)
    assert len(ops) == len(stmts)
    with open(os.path.join(GOLDEN, "manifest.json")) as fh:
        manifest = json.load(fh)
    assert manifest[name]["min_level"] == lo and manifest[name]["max_level"] == hi


def test_storage_names_follow_reference():
    gen, _, lo, hi = build_case("poisson2d_L8_v11_jacobi")
    st = gen.generate_storage(lo, hi, gen.finest_grid)
    assert [s.solution[0].to_exa() for s in st] == ["u@8", "u@7", "u@6"]
    assert [s.rhs[0].to_exa() for s in st] == ["RHS_u@8", "RHS_u@7", "RHS_u@6"]
    assert st[1].residual[0].to_exa() == "gen_residual_u@7"
    assert st[2].correction[0].to_exa() == "gen_error_u@6"
