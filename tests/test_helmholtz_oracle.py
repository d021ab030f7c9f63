"""Helmholtz (complex, indefinite) path of the oracle: the hand-written reference solver of
example_problems/Helmholtz/2D_FD_Helmholtz_fromL3.exa3 (V(2,1) RB-GS w=0.6 on the shifted operator M as
preconditioner of BiCGStab on A, BiCGStab coarse solver) restated as an op program."""
import numpy as np

from evostencils_amd import abi, candidates, problems
from oracle import oracle


def reference_vcycle_ops(lo, hi, omega=0.6, pre=2, post=1):
    """gen_mgCycle@(coarsest+1 .. finest) of exa3:202-272 as ops. On the finest level the cycle's unknown
    is `u` (SOL) and its right-hand side `f` (RHS)."""
    return candidates.v_cycle_ops(lo, hi, pre, post, omega=omega)


def test_reference_helmholtz_solver_converges():
    lo, hi = 3, 6
    k = 0.625 * (1 << hi)          # k h = 0.625 as in notebooks/helmholtz.ipynb:57
    pd = problems.helmholtz_2d(lo, hi, k=k)
    h = oracle.create(pd)
    out = h.evaluate(reference_vcycle_ops(lo, hi), 1, 0, 512)
    assert out.status == abi.SOLVE_CONVERGED
    assert out.final_res_norm < 1e-7 * out.init_res_norm
    # true residual of the returned solution against A u = f, computed independently with numpy
    u = h.get_field(hi, abi.F_CORR)
    n = (1 << hi) + 1
    hh = 1.0 / (n - 1)
    f = pd.right_hand_side(hi)
    Au = (4.0 / hh ** 2 - k ** 2) * u[1:-1, 1:-1] - (u[1:-1, :-2] + u[1:-1, 2:] + u[:-2, 1:-1] + u[2:, 1:-1]) / hh ** 2
    r = f[1:-1, 1:-1] - Au
    rel = np.sqrt(abs(np.sum(r * r))) / out.init_res_norm
    assert rel < 1e-6
    # Robin boundary: u_b = u_in / (1 - i k h) on the left and right edges
    assert np.allclose(u[1:-1, 0], u[1:-1, 1] / (1 - 1j * k * hh), rtol=1e-12, atol=0)
    assert np.allclose(u[1:-1, -1], u[1:-1, -2] / (1 - 1j * k * hh), rtol=1e-12, atol=0)
    assert np.all(u[0, :] == 0) and np.all(u[-1, :] == 0)


def test_wavenumber_retargeting_changes_the_operator():
    lo, hi = 3, 5
    pd = problems.helmholtz_2d(lo, hi, k=10.0)
    h = oracle.create(pd)
    a = h.evaluate(reference_vcycle_ops(lo, hi), 1, 0, 64)
    h.set_operators(pd.to_c({"k": 20.0}))
    b = h.evaluate(reference_vcycle_ops(lo, hi), 1, 0, 64)
    assert a.status == abi.SOLVE_CONVERGED and b.status == abi.SOLVE_CONVERGED
    assert a.init_res_norm == b.init_res_norm      # same right-hand side, zero initial guess
    assert not np.array_equal(h.get_field(hi, abi.F_CORR), np.zeros_like(h.get_field(hi, abi.F_CORR)))
    assert a.n_iterations != b.n_iterations or a.final_res_norm != b.final_res_norm


def evolved_double_cycle_ops(lo, hi, k):
    return candidates.evolved_helmholtz_double_cycle(lo, hi, k)


def test_evolved_double_cycle_fixture():
    lo, hi = 3, 6
    k = 0.625 * (1 << hi)
    gen, ops = evolved_double_cycle_ops(lo, hi, k)
    single = reference_vcycle_ops(lo, hi)
    drop_self_copies = lambda seq: [(o.opcode, o.level, o.partitioning, o.omega) for o in seq if not (o.opcode == abi.OP_COPY and o.dst == o.src0)]
    # statement for statement two hand-written cycles (prolongation weight: the grammar's 0.9999999999999999)
    assert [t[:3] for t in drop_self_copies(ops)] == [t[:3] for t in drop_self_copies(single)] * 2
    h = oracle.create(gen.problem)
    a, b = h.evaluate(single, 1, 0, 0), h.evaluate(ops, 1, 0, 512)
    assert a.status == b.status == abi.SOLVE_CONVERGED
    assert b.n_iterations < 0.6 * a.n_iterations           # 161 -> 88 (BiCGStab counts are sensitive to the last bit)
    u = h.get_field(hi, abi.F_CORR)
    n = (1 << hi) + 1
    hh = 1.0 / (n - 1)
    f = gen.problem.right_hand_side(hi)
    Au = (4.0 / hh ** 2 - k ** 2) * u[1:-1, 1:-1] - (u[1:-1, :-2] + u[1:-1, 2:] + u[:-2, 1:-1] + u[2:, 1:-1]) / hh ** 2
    r = f[1:-1, 1:-1] - Au
    assert np.sqrt(abs(np.sum(r * r))) / b.init_res_norm < 1e-6
