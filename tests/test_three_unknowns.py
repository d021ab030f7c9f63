"""Systems of THREE PDEs (round 6, ABI 6: EVOMG_MAX_UNKNOWNS 3). The reference's IR and lowering are general in the number of
unknowns (evostencils/ir/system.py:32-50, code_generation/exastencils.py:769-873); its only example of a system has two. CPU
half, on this repository's three-species problem (built-in `problems.three_species_2d` = the files under
tests/golden/problems/ThreeSpecies): the front end, the op programs, the oracle against an independent sparse direct solve, a
manufactured discrete solution, the problem's own solver settings. (The lowering's TEXT for two cycles on this problem is
pinned by goldens the reference's own lowering produced: tests/golden/threespecies2d_*.exa3, tests/test_lowering_golden.py.)"""
import os

import numpy as np
import pytest

from evostencils_amd import abi, problem_files, problems
from oracle import oracle

import cycles
from helpers import ops_for_case
from test_systems_oracle import apply_row

F = abi.field_of
FIXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "problems")
NF = 3


def v_cycle_ops(lo, hi, pre=2, post=1, omega=1.1, part=abi.PART_RED_BLACK, mask=0b111):
    """The generated solver's cycle on the three-unknown system: collective point smoother, per-unknown transfers."""
    ops = []
    sm = lambda L: abi.make_op(abi.OP_SMOOTH, L, F(abi.F_SOL), 0, F(abi.F_RHS), part, (1, 1, 1), omega, unknowns=mask)
    for L in range(hi, lo, -1):
        ops += [sm(L) for _ in range(pre)]
        ops += [abi.make_op(abi.OP_RESIDUAL, L, F(abi.F_RES, u), F(abi.F_SOL), F(abi.F_RHS, u)) for u in range(NF)]
        ops += [abi.make_op(abi.OP_RESTRICT, L - 1, F(abi.F_RHS, u), F(abi.F_RES, u)) for u in range(NF)]
        if L - 1 > lo:
            ops += [abi.make_op(abi.OP_SET_ZERO, L - 1, F(abi.F_SOL, u)) for u in range(NF)]
    for u in range(NF):
        ops.append(abi.make_op(abi.OP_COPY, lo, F(abi.F_CGS_RHS, u), F(abi.F_RHS, u)))
        ops.append(abi.make_op(abi.OP_SET_ZERO, lo, F(abi.F_CORR, u)))
    ops.append(abi.make_op(abi.OP_COARSE_SOLVE, lo))
    for L in range(lo + 1, hi + 1):
        ops += [abi.make_op(abi.OP_PROLONG_ADD, L, F(abi.F_SOL, u), F(abi.F_SOL, u), omega=1.0) for u in range(NF)]
        ops += [sm(L) for _ in range(post)]
    return ops


def direct_solution(pd, level):
    """The discrete system assembled as a sparse matrix over the inner nodes of all three unknowns and solved directly
    (scipy): independent of every loop of the oracle."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    n = (1 << level) + 1
    m = n - 2
    a = pd.system.assembled(level)
    u0, f = pd.system.initial_solution(level), pd.system.right_hand_side(level)
    idx = lambda u, i, j: u * m * m + (j - 1) * m + (i - 1)
    rows, cols, vals = [], [], []
    b = np.zeros(NF * m * m)
    for u in range(NF):
        for j in range(1, n - 1):
            for i in range(1, n - 1):
                r = idx(u, i, j)
                b[r] = f[u][j, i]
                for v in range(NF):
                    for e, (ox, oy) in enumerate(problems.NINE_POINT_OFFSETS):
                        c = a[u, v, e]
                        if c == 0.0:
                            continue
                        ii, jj = i + ox, j + oy
                        if 1 <= ii <= n - 2 and 1 <= jj <= n - 2:
                            rows.append(r); cols.append(idx(v, ii, jj)); vals.append(c)
                        else:
                            b[r] -= c * u0[v][jj, ii]                    # Dirichlet value
    x = spla.spsolve(sp.csr_matrix((vals, (rows, cols)), shape=(NF * m * m, NF * m * m)), b)
    out = []
    for u in range(NF):
        w = u0[u].copy()
        w[1:-1, 1:-1] = x[u * m * m:(u + 1) * m * m].reshape(m, m)
        out.append(w)
    return out


def test_the_problem_files_read_to_the_built_in_descriptor():
    p = problem_files.load_problem(FIXTURES, "ThreeSpecies/2D_FD_ThreeSpecies.settings", "ThreeSpecies/2D_FD_ThreeSpecies.knowledge")
    q = problems.three_species_2d(3, 6)
    assert (p.name, p.dimension, p.min_level, p.max_level) == (q.name, 2, 3, 6) and p.system.field_names == ["ca", "cb", "cc"]
    assert (p.max_iterations, p.cgs_max_iterations, p.target_res_reduction) == (q.max_iterations, q.cgs_max_iterations, q.target_res_reduction)
    for L in (3, 5, 6):
        assert np.array_equal(p.system.assembled(L), q.system.assembled(L))
    for a, b in zip(p.system.initial_solution(6) + p.system.right_hand_side(6), q.system.initial_solution(6) + q.system.right_hand_side(6)):
        assert np.allclose(a, b, rtol=1e-15, atol=1e-15)
    c = q.to_c()
    assert c.n_unknowns == 3 and c.abi_version == abi.ABI_VERSION == 6
    # the system operator is symmetric (CG is the coarse-grid solver): A[i][j] at offset o equals A[j][i] at -o
    a = q.system.assembled(5)
    mirror = [problems.NINE_POINT_OFFSETS.index((-ox, -oy)) for ox, oy in problems.NINE_POINT_OFFSETS]
    for i in range(NF):
        for j in range(NF):
            assert np.array_equal(a[i, j], a[j, i][mirror])


def test_op_programs_name_three_unknowns():
    gen, ops = ops_for_case("threespecies2d_L5_mixed")
    smooth = [(o.level, o.unknowns, tuple(o.block)[:2], o.partitioning) for o in ops if o.opcode == abi.OP_SMOOTH]
    # decoupled Jacobi: one statement per unknown; collective red-black: all three; collective blocks (2, 1): six cells
    assert smooth[:4] == [(5, 1, (1, 1), abi.PART_SINGLE), (5, 2, (1, 1), abi.PART_SINGLE), (5, 4, (1, 1), abi.PART_SINGLE),
                          (5, 7, (1, 1), abi.PART_RED_BLACK)]
    assert (4, 7, (2, 1), abi.PART_SINGLE) in smooth
    assert [o.dst // abi.F_COUNT for o in ops if o.opcode == abi.OP_RESIDUAL and o.level == 5] == [0, 1, 2]
    assert sum(1 for o in ops if o.opcode == abi.OP_COARSE_SOLVE) == 1
    # four unknowns are still one too many
    from evostencils_amd.code_generation import lowering
    with pytest.raises(lowering.UnsupportedExpression):
        lowering.Lowering([], 3, 5)._single_field([None] * 4)


@pytest.mark.parametrize("level", [4, 5])
def test_oracle_solves_the_discrete_system(level):
    pd = problems.three_species_2d(3, level)
    h = oracle.create(pd)
    pd.upload_initial_fields(h)
    out = h.evaluate(v_cycle_ops(3, level), 1, 0, 128)
    assert out.status == abi.SOLVE_CONVERGED and out.n_iterations <= 20
    assert out.history[-1] <= pd.target_res_reduction * out.history[0]
    rates = out.history[1:] / out.history[:-1]
    assert rates[1:].max() < 0.2                                    # V(2,1) collective red-black, w = 1.1
    want = direct_solution(pd, level)
    for u in range(NF):
        got = h.get_field(level, F(abi.F_SOL, u))
        assert np.abs(got - want[u]).max() <= 1e-9 * max(np.abs(want[u]).max(), 1e-300), u
    h.close()


def test_manufactured_discrete_solution_is_recovered():
    lo, hi = 3, 6
    pd = problems.three_species_2d(lo, hi)
    n = (1 << hi) + 1
    t = np.arange(n) / (n - 1)
    X, Y = t[None, :], t[:, None]
    w = [np.sin(2.0 * X) * np.cos(3.0 * Y) + X * Y, np.exp(X - 0.5 * Y) * (1.0 + X * X), np.cos(X + Y) * (1.0 - X * Y)]
    a = pd.system.assembled(hi)
    h = oracle.create(pd)
    for u in range(NF):
        f = np.zeros((n, n))
        f[1:-1, 1:-1] = apply_row(a, u, w)
        h.set_initial_field(hi, F(abi.F_SOL, u), np.where(problems._boundary_mask(n, 2), w[u], 0.0))
        h.set_initial_field(hi, F(abi.F_RHS, u), f)
    out = h.evaluate(v_cycle_ops(lo, hi), 1, 0, 128)
    assert out.status == abi.SOLVE_CONVERGED
    for u in range(NF):
        assert np.abs(h.get_field(hi, F(abi.F_SOL, u)) - w[u]).max() < 1e-8 * np.abs(w[u]).max()
    h.close()


@pytest.mark.parametrize("name", ["threespecies2d_L5_mixed", "threespecies2d_L6_v21_rb"])
@pytest.mark.parametrize("semantics", ["textbook", "exastencils-v1.1"])
def test_golden_cycles_run_on_the_oracle_in_both_semantics(name, semantics):
    gen, ops = ops_for_case(name)
    _, lo, hi, _ = cycles.CASES[name]
    h = oracle.create(problems.three_species_2d(lo, hi), jacobi_semantics=semantics)
    out = h.evaluate(ops, 1, 0, 128)
    assert out.status in (abi.SOLVE_CONVERGED, abi.SOLVE_MAXITER) and out.history[-1] < 1e-3 * out.history[0]
    if name.endswith("v21_rb"):
        assert out.status == abi.SOLVE_CONVERGED and out.conv_factor < 0.2       # red-black sweeps take effect in both
    h.close()
