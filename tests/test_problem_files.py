"""Problem-file front end: `.settings/.knowledge/.exa2/.exa3` -> ProblemDescription without an
ExaStencils run (the reference needs the compiler's debug layer-3 output, code_generation/parser.py:25-143).

Fixtures under tests/golden/problems/ are this repo's own problems (manufactured solutions), written in the same
layer-2 dialect; when the reference checkout is present its Poisson files must reproduce the built-in descriptors
bit for bit."""
import os

import numpy as np
import pytest

from evostencils_amd import abi, problem_files as pf, problems
from oracle import oracle

import cycles
from helpers import context_for
from evostencils_amd.code_generation.hip import ProgramGenerator

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = os.path.join(HERE, "golden", "problems")
REFERENCE = "/root/reference/example_problems"


def test_expression_evaluator_follows_python_precedence():
    env = {"PI": np.pi, "x": np.linspace(0.0, 1.0, 7), "h": 0.125}
    x, h = env["x"], env["h"]
    cases = {
        "PI**2 * cos ( PI * x ) - 4.0 * PI**2 * sin ( 2.0 * PI * x )":
            np.pi ** 2 * np.cos(np.pi * x) - 4.0 * np.pi ** 2 * np.sin(2.0 * np.pi * x),
        "2.0 / ( h ** 2 ) + 2.0 / ( h ** 2 )": 2.0 / (h ** 2) + 2.0 / (h ** 2),
        "-x ** 2": -x ** 2,
        "2 ** -1 ** 2": 2 ** -1 ** 2,
        "1.5e1 - 3 - 2": 1.5e1 - 3 - 2,
        "8 / 4 / 2 * 3": 8 / 4 / 2 * 3,
        "-(x - .5) * +x": -(x - .5) * +x,
        "sqrt(abs(x - 0.5)) + exp(-x)": np.sqrt(np.abs(x - 0.5)) + np.exp(-x),
    }
    for text, want in cases.items():
        got = pf.Expression(text).evaluate(env)
        assert np.array_equal(np.asarray(got), np.asarray(want)), text
    assert pf.Expression("a * (b + c)").names() == {"a", "b", "c"}


@pytest.mark.parametrize("text", ["1 +", "foo(2)", "2 $ 3", "(1 + 2", "1 2"])
def test_bad_expressions_are_reported(text):
    with pytest.raises(pf.UnsupportedProblem):
        pf.Expression(text).evaluate({})


def test_key_value_files_imports_comments_and_placeholders():
    k = pf.read_key_values(os.path.join(FIXTURES, "Diffusion", "2D_FD_AnisotropicDiffusion.knowledge"))
    assert k["dimensionality"] == "2" and k["maxLevel"] == "6"
    assert k["minLevel"] == "2"                       # the later assignment wins, the comment is dropped
    assert k["omp_numThreads"] == "4"                 # from the imported file
    s = pf.resolve_settings(pf.read_key_values(os.path.join(FIXTURES, "Diffusion", "2D_FD_AnisotropicDiffusion.settings")))
    assert s["l2file"] == "2D_FD_AnisotropicDiffusion.exa2"
    assert s["outputPath"] == "../generated/2D_FD_AnisotropicDiffusion/"


def _load(directory, name, **kw):
    return pf.load_problem(FIXTURES, f"{directory}/{name}.settings", f"{directory}/{name}.knowledge", **kw)


def test_descriptor_of_the_2d_fixture():
    p = _load("Diffusion", "2D_FD_AnisotropicDiffusion")
    assert (p.name, p.dimension, p.min_level, p.max_level) == ("2D_FD_AnisotropicDiffusion", 2, 2, 6)
    assert (p.field_name, p.rhs_name, p.operator_name, p.equation_name) == ("v", "load", "Diffusion", "diffusionEq")
    assert (p.max_iterations, p.cgs_max_iterations) == (60, 500)
    assert (p.target_res_reduction, p.cgs_target_res_reduction) == (1e-10, 1e-11)
    assert p.cgs_kind == abi.CGS_CG and p.scalar_kind == abi.REAL_F64 and p.bc_kind == abi.BC_DIRICHLET
    for level in (2, 6):
        h = 1.0 / (1 << level)
        assert p.mg_stencil(level, {}) == [0.2 / (h ** 2) + 2.0 / (h ** 2), -0.1 / (h ** 2), -0.1 / (h ** 2),
                                           -1.0 / (h ** 2), -1.0 / (h ** 2), 0.0, 0.0]
    u0 = p.initial_solution(4)
    x = np.arange(17) / 16.0
    assert np.array_equal(u0[0, :], x) and np.array_equal(u0[-1, :], x) and np.all(u0[:, -1] == 1.0)
    assert np.all(u0[1:-1, 1:-1] == 0.0)
    f = p.right_hand_side(4)
    assert f.shape == (17, 17) and f[8, 8] == 1.1 * np.pi ** 2 * np.sin(np.pi * 0.5) * np.sin(np.pi * 0.5)
    # what the grammar sees (parser.py:25-111): names and per-level stencils
    eqs, ops, fields = p.l2_information()
    assert [str(f) for f in fields] == ["v"] and eqs[0].name == "diffusionEq" and eqs[0].rhs_name == "load"
    assert {o.name for o in ops} == {"Diffusion", "gen_restrictionForRes_v", "gen_prolongationForSol_v"}


def test_descriptor_of_the_3d_fixture_level_selectors_and_domain():
    p = _load("Harmonic", "3D_FD_Harmonic")
    assert (p.dimension, p.min_level, p.max_level) == (3, 2, 4)
    assert p.domain_upper == (2.0, 1.0, 1.0)
    h = [2.0 / 16, 1.0 / 16, 1.0 / 16]
    assert p.mg_stencil(4, {}) == [2.0 / h[0] ** 2 + 2.0 / h[1] ** 2 + 2.0 / h[2] ** 2, -1.0 / h[0] ** 2, -1.0 / h[0] ** 2,
                                   -1.0 / h[1] ** 2, -1.0 / h[1] ** 2, -1.0 / h[2] ** 2, -1.0 / h[2] ** 2]
    u0 = p.initial_solution(4)                    # [i2][i1][i0]; boundary = x*y*z with x in (0,2)
    assert u0[16, 16, 16] == 2.0 and u0[16, 8, 16] == 2.0 * 0.5 * 1.0 and u0[5, 5, 5] == 0.0
    assert not p.right_hand_side(4).any()
    bumped = p.with_levels(3, 5)                  # the optimizer's problem-size bump keeps "finest" = the finest level
    assert bumped.initial_solution(5)[32, 32, 32] == 2.0


def _solve(p, smoother):
    gen = ProgramGenerator(problem=p, library=object())
    ctx = context_for(gen, p.min_level, p.max_level)
    depth = p.max_level - p.min_level
    expression, _ = cycles.v_cycle(ctx, depth, [smoother] * depth, [smoother] * depth, cgc_index=18)
    ops = gen.generate_op_program(expression, p.min_level, p.max_level)
    h = oracle.create(p)
    out = h.evaluate(ops, 1, 0, p.max_iterations)
    u = h.get_field(p.max_level, abi.F_SOL)
    h.close()
    return out, u


def test_2d_fixture_solves_to_the_manufactured_solution():
    p = _load("Diffusion", "2D_FD_AnisotropicDiffusion", min_level=3, max_level=6)
    out, u = _solve(p, [("rbgs", 18), ("rbgs", 18)])
    assert out.status == abi.OK and out.final_res_norm <= 1e-10 * out.init_res_norm
    x = np.arange(65) / 64.0
    exact = np.sin(np.pi * x)[None, :] * np.sin(np.pi * x)[:, None] + x[None, :]
    assert np.max(np.abs(u - exact)) < 5e-4              # second-order discretisation error at h = 1/64
    coarse, uc = _solve(_load("Diffusion", "2D_FD_AnisotropicDiffusion", min_level=3, max_level=5), [("rbgs", 18), ("rbgs", 18)])
    xc = np.arange(33) / 32.0
    exact_c = np.sin(np.pi * xc)[None, :] * np.sin(np.pi * xc)[:, None] + xc[None, :]
    ratio = np.max(np.abs(uc - exact_c)) / np.max(np.abs(u - exact))
    assert 3.5 < ratio < 4.5                             # O(h^2)


def test_3d_fixture_is_solved_exactly():
    p = _load("Harmonic", "3D_FD_Harmonic")
    out, u = _solve(p, [("rbgs", 18)])
    assert out.status == abi.OK and out.n_iterations < p.max_iterations
    n = 17
    x = 2.0 * np.arange(n) / 16.0
    y = np.arange(n) / 16.0
    exact = x[None, None, :] * y[None, :, None] * y[:, None, None]
    assert np.max(np.abs(u - exact)) < 1e-11             # trilinear: no discretisation error


@pytest.mark.parametrize("edit, message", [
    (lambda s: s.replace("[ 1,  0] => -0.1", "[ 2,  0] => -0.1"), "outside the star stencil"),      # (a corner entry is fine: nine-point stencils run as a system of one unknown)
    (lambda s: s.replace("with Real on Node", "with Real on Cell", 1), "only `Real on Node`"),
    (lambda s: s.replace("Diffusion * v == load", "Diffusion * v + v == load"), "not a scaled stencil"),
    (lambda s: s.replace("-0.1 / ( vf_gridWidth_x ** 2 )", "-0.1 * vf_nodePos_x / ( vf_gridWidth_x ** 2 )", 1), "constant coefficients"),
    (lambda s: s + "\nEquation second {\n  Diffusion * v == load\n}\n", "do not match"),
    (lambda s: s + "\nGlobals {\n  k is 1.0\n}\n", "Globals entry not understood"),
])
def test_unsupported_constructs_are_named(edit, message):
    with open(os.path.join(FIXTURES, "Diffusion", "2D_FD_AnisotropicDiffusion.exa2")) as fh:
        text = edit(fh.read())
    with pytest.raises(pf.UnsupportedProblem, match=message):
        pf.describe(pf.parse_layer2(text), "edited", 2, 2, 5, {})


def test_program_generator_takes_problem_directories():
    gen = ProgramGenerator("unused.jar", FIXTURES, "Diffusion/2D_FD_AnisotropicDiffusion.settings",
                           "Diffusion/2D_FD_AnisotropicDiffusion.knowledge", "unused.platform")
    assert gen.problem_name == "2D_FD_AnisotropicDiffusion" and gen.dimension == 2
    assert (gen.min_level, gen.max_level) == (2, 6)
    assert [str(f) for f in gen.fields] == ["v"]
    assert gen.finest_grid[0].spacing == (1.0 / 64, 1.0 / 64)
    gen3 = ProgramGenerator("unused.jar", FIXTURES, "Harmonic/3D_FD_Harmonic.settings", "Harmonic/3D_FD_Harmonic.knowledge")
    assert gen3.finest_grid[0].spacing == (2.0 / 16, 1.0 / 16, 1.0 / 16)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present")
@pytest.mark.parametrize("name, builtin", [("2D_FD_Poisson_fromL2", problems.poisson_2d), ("3D_FD_Poisson_fromL2", problems.poisson_3d)])
def test_reference_poisson_files_reproduce_the_builtin_descriptors(name, builtin):
    p = pf.load_problem(REFERENCE, f"Poisson/{name}.settings", f"Poisson/{name}.knowledge")
    b = builtin()
    assert (p.name, p.dimension, p.min_level, p.max_level) == (b.name, b.dimension, b.min_level, b.max_level)
    for attr in ("field_name", "rhs_name", "operator_name", "equation_name", "restriction_name", "prolongation_name",
                 "max_iterations", "cgs_max_iterations", "target_res_reduction", "cgs_target_res_reduction", "cgs_kind",
                 "scalar_kind", "bc_kind", "outer_kind"):
        assert getattr(p, attr) == getattr(b, attr), attr
    for level in range(b.min_level, b.max_level + 1):
        assert p.mg_stencil(level, {}) == b.mg_stencil(level, {})
    L = b.max_level
    assert np.array_equal(p.initial_solution(L), b.initial_solution(L))
    assert np.array_equal(p.right_hand_side(L), b.right_hand_side(L))


# ---- systems of PDEs from problem files (SURVEY 8f N2 x N3) -----------------------------------------------------------
def test_system_descriptor_from_files():
    """Two unknowns, equations as sums of scaled named stencils with `Globals` in the coefficients, a four-corner cross
    derivative, the unknown order taken from the layer-3 `generate solver for ux in momentumX and uy in momentumY`."""
    p = pf.load_problem(FIXTURES, "Elasticity/2D_FD_Elasticity.settings", "Elasticity/2D_FD_Elasticity.knowledge")
    spec = p.system
    assert (p.name, p.dimension, p.min_level, p.max_level) == ("2D_FD_Elasticity", 2, 3, 6)
    assert spec.field_names == ["ux", "uy"] and spec.rhs_names == ["fx", "fy"] and spec.equation_names == ["momentumX", "momentumY"]
    assert p.parameters == {"firstLame": 1.5, "shear": 1.0}
    assert (p.max_iterations, p.cgs_max_iterations, p.target_res_reduction) == (80, 800, 1e-11)
    a = spec.assembled(5)                          # [equation][unknown][centre, W, E, S, N, SW, SE, NW, NE], h = 1/32
    s = 1024.0
    assert np.allclose(a[0, 0], [-2 * 3.5 * s - 2 * s, 3.5 * s, 3.5 * s, s, s, 0, 0, 0, 0])
    assert np.allclose(a[1, 1], [-2 * s - 2 * 3.5 * s, s, s, 3.5 * s, 3.5 * s, 0, 0, 0, 0])
    assert np.allclose(a[0, 1], [0, 0, 0, 0, 0, 2.5 * 0.25 * s, -2.5 * 0.25 * s, -2.5 * 0.25 * s, 2.5 * 0.25 * s])
    assert np.array_equal(a[0, 1], a[1, 0])
    u0 = spec.initial_solution(4)
    x = np.arange(17) / 16.0
    assert np.all(u0[0] == 0.0) and np.all(u0[1][1:-1, 1:-1] == 0.0)
    assert np.allclose(u0[1][-1, :], 0.25 * x * (1 - x)) and np.all(u0[1][0, :] == 0.0)
    f = spec.right_hand_side(4)
    assert np.all(f[0] == 0.0) and np.allclose(f[1], 1.5e-3 * np.sin(np.pi * x)[None, :] * np.ones((17, 1)))
    c = p.to_c()
    assert c.n_unknowns == 2 and c.sys_stencil[5][0][1][5] == a[0, 1, 5]


def test_program_generator_on_a_system_directory():
    gen = ProgramGenerator("unused.jar", FIXTURES, "Elasticity/2D_FD_Elasticity.settings", "Elasticity/2D_FD_Elasticity.knowledge")
    assert [str(f) for f in gen.fields] == ["ux", "uy"] and (gen.min_level, gen.max_level) == (3, 6)
    ctx = context_for(gen, 3, 6)
    expression, _ = cycles.v_cycle(ctx, 3, [[("rbgs", 18)]] * 3, [[("decoupled", 14)]] * 3)
    text = gen.generate_cycle_function(expression, gen.generate_storage(3, 6, gen.finest_grid), 3, 6, 6)
    assert "gen_residual_ux@6 = fx@6 - " in text.replace("(", "").replace(")", "") or "gen_residual_ux@6" in text
    ops = gen.generate_op_program(expression, 3, 6)
    assert any(o.opcode == abi.OP_SMOOTH and o.unknowns == 3 for o in ops) and any(o.opcode == abi.OP_SMOOTH and o.unknowns in (1, 2) for o in ops)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present")
def test_reference_elasticity_files_reproduce_the_builtin_descriptor():
    p = pf.load_problem(REFERENCE, "LinearElasticity/2D_FD_LinearElasticity_fromL2.settings",
                        "LinearElasticity/2D_FD_LinearElasticity_fromL2.knowledge")
    b = problems.linear_elasticity_2d()
    assert (p.name, p.min_level, p.max_level, p.parameters) == (b.name, b.min_level, b.max_level, b.parameters)
    for attr in ("field_names", "rhs_names", "equation_names", "restriction_names", "prolongation_names"):
        assert getattr(p.system, attr) == getattr(b.system, attr), attr
    for level in range(b.min_level, b.max_level + 1):
        assert np.array_equal(p.system.assembled(level), b.system.assembled(level))
    L = b.max_level
    for x, y in zip(p.system.initial_solution(L) + p.system.right_hand_side(L), b.system.initial_solution(L) + b.system.right_hand_side(L)):
        assert np.array_equal(x, y)


def test_layer3_helmholtz_type_problem_is_read_from_its_files():
    """A Helmholtz-type problem enters at layer 3 (no .exa2): complex field, outer and preconditioner operators over
    Globals, Robin / Dirichlet boundaries on layer 4, hand-written Krylov solvers whose caps and tolerances are read. The
    fixture is this repository's own problem (tests/golden/problems/Waveguide) with numbers that differ from the
    reference's example everywhere."""
    p = _load("Waveguide", "2D_FD_Waveguide_fromL3")
    assert (p.name, p.dimension, p.min_level, p.max_level) == ("2D_FD_Waveguide_fromL3", 2, 3, 6)
    assert (p.field_name, p.rhs_name, p.operator_name, p.equation_name) == ("w", "g", "Damped", "DampedEq")
    assert (p.restriction_name, p.prolongation_name) == ("Restrict_w", "Prolong_w")
    assert p.scalar_kind == abi.COMPLEX_F64 and p.bc_kind == abi.BC_HELMHOLTZ_ROBIN
    assert p.outer_kind == abi.OUTER_PBICGSTAB and p.cgs_kind == abi.CGS_BICGSTAB
    assert (p.max_iterations, p.target_res_reduction, p.cgs_max_iterations, p.cgs_target_res_reduction) == (400, 1e-8, 300, 1e-5)
    assert p.parameters == {"k": 24.0}
    for level in (3, 6):
        h = 1.0 / (1 << level)
        o = -1.0 / (h ** 2)
        assert p.mg_stencil(level, {"k": 24.0}) == [2.0 / (h ** 2) + 2.0 / (h ** 2) - 24.0 ** 2 * (1.0 + 0.7j), o, o, o, o, 0.0, 0.0]
        assert p.outer_stencil(level, {"k": 48.0}) == [2.0 / (h ** 2) + 2.0 / (h ** 2) - 48.0 ** 2, o, o, o, o, 0.0, 0.0]
    f = p.right_hand_side(5)
    x = np.arange(33) / 32.0
    want = np.exp(-60.0 * ((x[None, :] - 0.25) ** 2 + (x[:, None] - 0.5) ** 2))
    assert f.dtype == np.complex128 and np.array_equal(f, want.astype(np.complex128))
    assert not p.initial_solution(4).any()


def test_layer3_problem_solves_through_the_oracle():
    """files -> ProgramGenerator -> op program -> oracle: the preconditioned BiCGStab converges to the file's tolerance and
    the returned field satisfies the file's equation and Robin condition."""
    gen = ProgramGenerator(None, FIXTURES, "Waveguide/2D_FD_Waveguide_fromL3.settings", "Waveguide/2D_FD_Waveguide_fromL3.knowledge",
                           library=object())
    p = gen.problem
    assert p.parameters == {"k": 24.0} and p.max_iterations == 400
    from test_helmholtz_oracle import reference_vcycle_ops
    ops = reference_vcycle_ops(p.min_level, p.max_level)
    h = oracle.create(p)
    out = h.evaluate(ops, 1, 0, 512)
    assert out.status == abi.SOLVE_CONVERGED and out.final_res_norm < 1e-8 * out.init_res_norm
    u = h.get_field(p.max_level, abi.F_CORR)
    n = (1 << p.max_level) + 1
    hh = 1.0 / (n - 1)
    f = p.right_hand_side(p.max_level)
    a = (4.0 / hh ** 2 - 24.0 ** 2) * u[1:-1, 1:-1] - (u[1:-1, :-2] + u[1:-1, 2:] + u[:-2, 1:-1] + u[2:, 1:-1]) / hh ** 2
    assert np.abs(a - f[1:-1, 1:-1]).max() <= 1e-6 * np.abs(f).max()
    assert np.abs(u[1:-1, 0] - u[1:-1, 1] / (1.0 - 1j * 24.0 * hh)).max() <= 1e-12 * np.abs(u).max()
    h.close()


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present")
def test_reference_helmholtz_files_reproduce_the_builtin_descriptor():
    """The reference's Helmholtz directory (layer-3 entry) parses to exactly problems.helmholtz_2d."""
    p = pf.load_problem(REFERENCE, "Helmholtz/2D_FD_Helmholtz_fromL3.settings", "Helmholtz/2D_FD_Helmholtz_fromL3.knowledge")
    b = problems.helmholtz_2d(3, 7)
    for attr in ("name", "dimension", "min_level", "max_level", "scalar_kind", "bc_kind", "outer_kind", "cgs_kind", "max_iterations",
                 "cgs_max_iterations", "target_res_reduction", "cgs_target_res_reduction", "field_name", "rhs_name", "operator_name",
                 "equation_name", "restriction_name", "prolongation_name", "parameters"):
        assert getattr(p, attr) == getattr(b, attr), attr
    for level in (3, 5, 7, 11):
        for k in (80.0, 160.0, 1280.0):
            assert p.mg_stencil(level, {"k": k}) == b.mg_stencil(level, {"k": k})
            assert p.outer_stencil(level, {"k": k}) == b.outer_stencil(level, {"k": k})
    for level in (3, 7):
        assert np.array_equal(p.right_hand_side(level), b.right_hand_side(level))
        assert np.array_equal(p.initial_solution(level), b.initial_solution(level))


def test_scalar_problem_with_a_nine_point_stencil_becomes_a_system_of_one_unknown():
    """code_generation/parser.py:8-22 reads stencils with any offsets; a 2D scalar operator with corner entries (the compact
    "Mehrstellen" Laplacian of tests/golden/problems/Mehrstellen) is described as a system of ONE unknown — descriptor
    stencil_kind = STENCIL_COMPACT, nine coefficients per level — and solved by the oracle to the discretisation error of its
    manufactured solution; a stencil that reaches beyond the 3 x 3 neighbourhood is still rejected."""
    from evostencils_amd import abi, candidates
    from evostencils_amd.code_generation.hip import ProgramGenerator
    from oracle import oracle
    p = _load("Mehrstellen", "2D_FD_Mehrstellen")
    assert p.system is not None and p.system.field_names == ["w"] and p.system.equation_names == ["compactEq"]
    c = p.to_c()
    assert (c.n_unknowns, c.stencil_kind) == (1, abi.STENCIL_COMPACT)
    h2 = (1.0 / (1 << 4)) ** 2
    want = np.array([20.0, -4.0, -4.0, -4.0, -4.0, -1.0, -1.0, -1.0, -1.0]) / (6.0 * h2)
    assert np.allclose([c.sys_stencil[4][0][0][e] for e in range(9)], want, rtol=1e-15)
    gen = ProgramGenerator(None, FIXTURES, "Mehrstellen/2D_FD_Mehrstellen.settings", "Mehrstellen/2D_FD_Mehrstellen.knowledge",
                           library=object(), jacobi_semantics="textbook")
    lo, hi = gen.min_level, gen.max_level
    rb = [("rbgs", 18)]
    expression = candidates.v_cycle(candidates.production_context(gen, lo, hi), hi - lo, [rb * 2] * (hi - lo), [rb] * (hi - lo))[0]
    ops = gen.generate_op_program(expression, lo, hi)
    assert all(o.unknowns == 0 for o in ops)                  # a scalar op program: the backend supplies the mask of the one unknown
    handle = oracle.create(gen.problem)
    out = handle.evaluate(ops, 1, 0, 64)
    assert out.status == abi.SOLVE_CONVERGED and out.n_iterations <= 12
    n = (1 << hi) + 1
    x = np.arange(n) / (n - 1)
    exact = np.sin(np.pi * x)[None, :] * np.sin(np.pi * x)[:, None] + x[None, :]
    assert np.abs(handle.get_field(hi, abi.F_SOL) - exact).max() < 1e-3
    # a candidate whose operator is NOT the problem's is refused at lowering (the op program carries no coefficients)
    other = ProgramGenerator(None, FIXTURES, "Diffusion/2D_FD_AnisotropicDiffusion.settings", "Diffusion/2D_FD_AnisotropicDiffusion.knowledge",
                             library=object(), jacobi_semantics="textbook")
    other.reinitialize(lo, hi, {})
    foreign = candidates.v_cycle(candidates.production_context(other, lo, hi), hi - lo, [rb] * (hi - lo), [rb] * (hi - lo))[0]
    from evostencils_amd.code_generation import lowering
    with pytest.raises(lowering.UnsupportedExpression):
        gen.generate_op_program(foreign, lo, hi)
    wide = pf.parse_layer2(open(os.path.join(FIXTURES, "Mehrstellen", "2D_FD_Mehrstellen.exa2")).read().replace("[-1, -1]", "[-2, 0]"))
    with pytest.raises(pf.UnsupportedProblem):
        pf.describe(wide, "wide", 2, 2, 5)
