"""Problems read from `.settings/.knowledge/.exa2/.exa3` files (evostencils_amd/problem_files.py) solved on the GPU
through the reference-style constructor, against the CPU oracle: anisotropic coefficients and a non-cubic mesh width
are operator values none of the built-in descriptors exercise."""
import os

import numpy as np
import pytest

from evostencils_amd import abi
from evostencils_amd.code_generation.hip import ProgramGenerator

import cycles
from helpers import context_for

pytestmark = pytest.mark.gpu

FIXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "problems")
REL_TOL = 1e-10

CASES = {
    "diffusion2d": ("Diffusion/2D_FD_AnisotropicDiffusion", [("rbgs", 18), ("jacobi", 14)], 3, 7),
    "diffusion2d_block": ("Diffusion/2D_FD_AnisotropicDiffusion", [("block", 16, (1, 4))], 3, 6),
    "harmonic3d": ("Harmonic/3D_FD_Harmonic", [("rbgs", 18)], 2, 5),
    "harmonic3d_block": ("Harmonic/3D_FD_Harmonic", [("block", 12, (2, 1, 1)), ("jacobi", 12)], 2, 4),
    # a system of two PDEs read from files (`Globals`, equations as sums of scaled stencils, four-corner cross derivative)
    "elasticity2d": ("Elasticity/2D_FD_Elasticity", [("rbgs", 18)], 3, 6),
    "elasticity2d_decoupled": ("Elasticity/2D_FD_Elasticity", [("decoupled", 14), ("blocks", 14, ((2, 1), (2, 1)))], 3, 5),
    # a SCALAR problem whose stencil is not a star (compact nine-point Laplacian: code_generation/parser.py:8-22 admits any
    # offsets): it runs as a system of one unknown (EVOMG_STENCIL_COMPACT)
    "mehrstellen2d": ("Mehrstellen/2D_FD_Mehrstellen", [("rbgs", 18)], 2, 6),
    "mehrstellen2d_jacobi_block": ("Mehrstellen/2D_FD_Mehrstellen", [("jacobi", 14), ("block", 14, (2, 2))], 3, 7),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_file_defined_problem_matches_oracle(case):
    from oracle import oracle
    stem, smoother, lo, hi = CASES[case]
    gen = ProgramGenerator("unused.jar", FIXTURES, stem + ".settings", stem + ".knowledge", "unused.platform",
                           jacobi_semantics="textbook")
    gen.reinitialize(lo, hi, {})
    pd = gen.problem
    assert (pd.min_level, pd.max_level) == (lo, hi)
    ctx = context_for(gen, lo, hi)
    depth = hi - lo
    expression, _ = cycles.v_cycle(ctx, depth, [smoother] * depth, [smoother] * depth, cgc_index=18)
    storages = gen.generate_storage(lo, hi, gen.finest_grid)
    gen.initialize_code_generation(lo, hi)
    t_ms, rho, iterations = gen.generate_and_evaluate(expression, storages, lo, hi, "", evaluation_samples=1)
    ops = gen.generate_op_program(expression, lo, hi)
    ho = oracle.create(pd)
    want = ho.evaluate(ops, 1, 0, pd.max_iterations + 2)
    _, want_rho, want_iterations = ProgramGenerator.interpret_outcome(want, 1e100)
    assert want.status == abi.OK and iterations == want_iterations and t_ms > 0
    assert rho == pytest.approx(want_rho, rel=1e-10)
    handle = gen._ensure_handle(lo, hi, None)
    got = handle.evaluate(ops, 1, 0, pd.max_iterations + 2)
    assert abs(got.final_res_norm - want.final_res_norm) <= REL_TOL * want.final_res_norm + 1e-300
    ug = handle.get_field(hi, abi.F_SOL)
    uo = ho.get_field(hi, abi.F_SOL)
    assert np.max(np.abs(ug - uo)) <= 1e-10 * np.max(np.abs(uo))
    ho.close()
    gen.close()


def test_layer3_helmholtz_type_problem_from_files_matches_oracle():
    """The repository's own wave-guide problem (layer-3 / layer-4 entry, tests/golden/problems/Waveguide) through the
    reference-style constructor on the GPU: same outer iteration count as the oracle, history and solution within 1e-10."""
    from oracle import oracle
    from test_helmholtz_oracle import reference_vcycle_ops
    stem = "Waveguide/2D_FD_Waveguide_fromL3"
    gen = ProgramGenerator("unused.jar", FIXTURES, stem + ".settings", stem + ".knowledge", "unused.platform")
    pd = gen.problem
    assert pd.parameters == {"k": 24.0} and (pd.min_level, pd.max_level) == (3, 6)
    ops = reference_vcycle_ops(pd.min_level, pd.max_level)
    handle = gen._ensure_handle(pd.min_level, pd.max_level, None)
    got = handle.evaluate(ops, 1, 0, 512)
    ho = oracle.create(pd)
    want = ho.evaluate(ops, 1, 0, 512)
    assert got.status == want.status == abi.SOLVE_CONVERGED and got.n_iterations == want.n_iterations
    assert (np.abs(got.history - want.history) / want.history).max() <= REL_TOL
    a, b = handle.get_field(pd.max_level, abi.F_CORR), ho.get_field(pd.max_level, abi.F_CORR)
    assert np.abs(a - b).max() <= REL_TOL * np.abs(b).max()
    ho.close()
    gen.close()
