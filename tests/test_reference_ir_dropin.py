"""Drop-in check on the IR side (runs only where the read-only reference checkout exists, i.e. in the build
container; skipped on the GPU box): DAGs built from the REFERENCE's own evostencils.ir classes by the
reference's own grammar productions are accepted by our ProgramGenerator unchanged — same ExaSlang text as the
golden vectors, same op program as for our own IR."""
import os
import sys

import pytest

import cycles
from helpers import GOLDEN, build_case, generator_for

REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "evostencils")),
                                reason="reference checkout not present")


@pytest.fixture(scope="module")
def golden_module():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import generate_golden as gg
    gg.install_deap_stub()
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    yield gg
    for name in [m for m in sys.modules if m == "deap" or m.startswith("deap.") or m == "evostencils" or m.startswith("evostencils.")]:
        del sys.modules[name]
    sys.path.remove(REFERENCE)


@pytest.mark.parametrize("name", sorted(cycles.CASES))
def test_reference_dags_lower_through_our_backend(golden_module, name):
    problem, lo, hi, builder = cycles.CASES[name]
    ref_ctx, _ref_gen, _ref_storages = golden_module.reference_context(problem, lo, hi)
    ref_expression, _ = builder(ref_ctx)
    assert type(ref_expression).__module__.startswith("evostencils.ir")     # really the reference's classes
    ours = generator_for(problem, lo, hi)
    text = ours.generate_cycle_function(ref_expression, None, lo, hi, hi)
    with open(os.path.join(GOLDEN, f"{name}.exa3")) as fh:
        assert text == fh.read()
    ops_from_reference_ir = ours.generate_op_program(ref_expression, lo, hi)
    _gen, own_expression, _, _ = build_case(name)
    ops_from_own_ir = ours.generate_op_program(own_expression, lo, hi)
    assert [o.key() for o in ops_from_reference_ir] == [o.key() for o in ops_from_own_ir]
