"""Random individuals of the reference's grammar (tests/golden/grammar_individuals.json: strings grown over the reference's
own typed primitive sets in the build container and accepted by the reference's lowering — generator script committed
beside them) compiled with this repository's productions and solved on the GPU against the CPU oracle: arbitrary nesting
(several coarse-grid corrections, smoothers between coarsening steps, every block shape the grammar registers — for the
two-unknown problem mostly one shape per unknown), in the reference pipeline's `with jacobi` semantics (the drop-in
default) and the textbook one. Same status, same iteration count, residual history within 1e-10."""
import json
import os

import numpy as np
import pytest

from evostencils_amd import abi
from evostencils_amd.code_generation.hip import ProgramGenerator
from evostencils_amd.grammar import productions
from helpers import GOLDEN, context_for

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "grammar_individuals.json")) as _fh:
    INDIVIDUALS = json.load(_fh)
# finest level per problem: grids on which the strip / tile kernels and the interpreter's fusions are active
LEVELS = {"2D_FD_Poisson_fromL2": 8, "3D_FD_Poisson_fromL2": 6, "2D_FD_LinearElasticity_fromL2": 7}


@pytest.mark.parametrize("semantics", ["exastencils-v1.1", "textbook"])
@pytest.mark.parametrize("problem", sorted(INDIVIDUALS))
def test_grammar_individuals_match_the_oracle(problem, semantics, monkeypatch):
    from evostencils_amd import native
    from oracle import oracle
    entry = INDIVIDUALS[problem]
    hi = LEVELS[problem]
    lo = hi - entry["depth"]
    if problem.startswith("3D"):
        monkeypatch.setenv("EVOMG_STRIP3D_MIN_N", "17")          # the 3D tiles and their fusions on 65^3 / 33^3 / 17^3
    import dataclasses
    from evostencils_amd import problems
    # twelve iterations per individual are plenty to compare trajectories (most random individuals are poor solvers and
    # would run into the 100-iteration cap)
    pd = dataclasses.replace(problems.BY_NAME[problem](lo, hi), max_iterations=12)
    gen = ProgramGenerator(problem=pd, library=native.library(), jacobi_semantics=semantics)
    gen.initialize_code_generation(lo, hi)
    ho = oracle.create(gen.problem)
    compared = converged = rejected = 0
    for text in entry["individuals"]:
        state = productions.compile_individual(text, context_for(gen, lo, hi))
        ops = gen.generate_op_program(state[0], lo, hi)
        rc_got, got = gen._handle.try_evaluate(ops, 1, 0, 128)
        rc_want, want = ho.try_evaluate(ops, 1, 0, 128)
        assert rc_got == rc_want, text
        if rc_want == abi.ERR_UNSUPPORTED:
            # a block smoother with one shape per unknown where its result would be used (textbook semantics): rejected by
            # both sides (tests/test_systems_oracle.py::test_mixed_shape_blocks_only_where_their_result_is_never_used)
            assert semantics == "textbook" and any(any(o.block_other) for o in ops), text
            rejected += 1
            continue
        assert rc_want == abi.OK, text
        assert got.status == want.status and got.n_iterations == want.n_iterations, text
        finite = np.isfinite(want.history) & (want.history > 1e-13 * want.history[0])
        assert np.array_equal(np.isfinite(got.history), np.isfinite(want.history)), text
        rel = np.abs(got.history[finite] - want.history[finite]) / want.history[finite]
        assert rel.max() <= 1e-10, (text, rel.max())
        compared += 1
        converged += want.status == abi.SOLVE_CONVERGED
    assert compared + rejected == len(entry["individuals"]) and compared >= rejected
    ho.close()
    gen.close()
