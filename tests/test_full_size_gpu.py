"""BASELINE.json's full-size configurations on the GPU (the oracle cannot finish these in seconds): checked through
size-independent properties — known analytic solutions, textbook convergence factors, residual reduction reached,
consistency of the Robin boundary — plus repeatability (two evaluations give identical histories)."""
import numpy as np
import pytest

from evostencils_amd import abi, candidates, problems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()
    return nat


def vcycle(lo, hi, pre, post, smoother):
    return candidates.v_cycle_ops(lo, hi, pre, post, smoother)


def test_c2_poisson2d_4096_six_level_rbgs(native):
    """configs[1]: 2D Poisson 4096^2, 6-level V(1,1) RB-GS."""
    lo, hi = 7, 12
    pd = problems.poisson_2d(lo, hi)
    h = native.create(pd)
    ops = vcycle(lo, hi, 1, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.0))
    a = h.evaluate(ops, 1, 0, 128)
    assert a.status == abi.SOLVE_CONVERGED and a.conv_factor < 0.1 and a.n_iterations <= 14
    assert a.final_res_norm <= 1e-12 * a.init_res_norm
    n = (1 << hi) + 1
    x = np.arange(n) / (n - 1)
    exact = np.cos(np.pi * x[None, :]) - np.sin(2 * np.pi * x[:, None])
    err = np.abs(h.get_field(hi, abi.F_SOL) - exact).max()
    hh = 1.0 / (n - 1)
    assert err < 5.0 * hh * hh * np.pi ** 2          # O(h^2) discretisation error (measured 7.4e-4 at h = 1/64)
    b = h.evaluate(ops, 1, 0, 128)
    assert np.array_equal(a.history, b.history)


def test_c2_full_size_against_the_oracle(native):
    """The headline configuration at FULL size against the CPU oracle (one oracle solve of 4097^2 takes a couple of
    seconds on the host cores): same iteration count, residual history within 1e-10 relative, final solution
    bit-identical."""
    from oracle import oracle
    lo, hi = 7, 12
    pd = problems.poisson_2d(lo, hi)
    ops = vcycle(lo, hi, 1, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.0))
    hg = native.create(pd)
    got = hg.evaluate(ops, 1, 0, 128)
    ug = hg.get_field(hi, abi.F_SOL)
    hg.close()
    ho = oracle.create(pd)
    want = ho.evaluate(ops, 1, 0, 128)
    uo = ho.get_field(hi, abi.F_SOL)
    ho.close()
    assert got.status == want.status == abi.SOLVE_CONVERGED
    assert got.n_iterations == want.n_iterations
    assert len(got.history) == len(want.history)
    assert np.all(np.abs(got.history - want.history) <= 1e-10 * want.history)
    assert np.array_equal(ug, uo)


def test_c4_poisson3d_512_v22_block_jacobi(native):
    """configs[3]: 3D Poisson 512^3, V(2,2) block-Jacobi (2,1,1), w = 0.8, 5 levels."""
    lo, hi = 5, 9
    pd = problems.poisson_3d(lo, hi)
    h = native.create(pd)
    ops = vcycle(lo, hi, 2, 2, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 1, 1), 0.8))
    out = h.evaluate(ops, 1, 0, 128)
    assert out.status == abi.SOLVE_CONVERGED and out.conv_factor < 0.5
    n = (1 << hi) + 1
    x = np.arange(n) / (n - 1)
    got = h.get_field(hi, abi.F_SOL)
    # harmonic quadratic boundary data: the 7-point stencil reproduces it exactly (3D_FD_Poisson_fromL2.exa2:6-9)
    worst = 0.0
    for k in range(0, n, 64):
        exact = x[None, :] ** 2 - 0.5 * x[:, None] ** 2 - 0.5 * x[k] ** 2
        worst = max(worst, np.abs(got[k] - exact).max())
    assert worst < 1e-9
    print(f"C4: {out.n_iterations:.0f} iterations, rho {out.conv_factor:.4f}, {out.time_ms:.1f} ms")


def test_c3_helmholtz_2048_reference_cycle(native):
    """configs[2]: 2D Helmholtz 2048^2 (complex), k = 1280 (k h = 0.625), MG-preconditioned BiCGStab with the reference
    V(2,1) RB-GS w = 0.6 cycle (Helmholtz exa3:202-272), 5 levels. With the reference cycle the outer iteration count
    grows ~3.3x per level (oracle: 637 / 2052 / 6986 iterations at levels 7 / 8 / 9), i.e. beyond the reference's own
    cap of 10000 at level 11 — that is what evolved cycles are for. So the full-size run is capped at 150 outer
    iterations and checked through invariants: the recurrence residual equals the true residual f - A u of the
    returned iterate, the Robin/Dirichlet boundary conditions hold, and the run is reproducible bit for bit."""
    import dataclasses
    from test_helmholtz_oracle import reference_vcycle_ops
    lo, hi = 7, 11
    k = 0.625 * (1 << hi)
    pd = dataclasses.replace(problems.helmholtz_2d(lo, hi, k=k), max_iterations=150)
    h = native.create(pd)
    ops = reference_vcycle_ops(lo, hi)
    out = h.evaluate(ops, 1, 600e3, 160)
    assert out.status == abi.SOLVE_MAXITER and out.n_iterations == 150
    assert np.all(np.isfinite(out.history)) and len(out.history) == 151
    u = h.get_field(hi, abi.F_CORR)
    n = (1 << hi) + 1
    hh = 1.0 / (n - 1)
    assert np.allclose(u[1:-1, 0], u[1:-1, 1] / (1 - 1j * k * hh), rtol=1e-12, atol=0)
    assert np.allclose(u[1:-1, -1], u[1:-1, -2] / (1 - 1j * k * hh), rtol=1e-12, atol=0)
    assert np.all(u[0, :] == 0) and np.all(u[-1, :] == 0)
    f = pd.right_hand_side(hi)
    Au = (4.0 / hh ** 2 - k ** 2) * u[1:-1, 1:-1] - (u[1:-1, :-2] + u[1:-1, 2:] + u[:-2, 1:-1] + u[2:, 1:-1]) / hh ** 2
    r = f[1:-1, 1:-1] - Au
    true_res = np.sqrt(abs(np.sum(r * r)))
    assert abs(true_res - out.final_res_norm) <= 1e-6 * out.init_res_norm
    again = h.evaluate(ops, 1, 600e3, 160)
    assert np.array_equal(again.history, out.history)
    print(f"C3: 150 outer iterations in {out.time_ms:.0f} ms ({out.time_ms / 150:.2f} ms per iteration = 2 MG cycles + 5 dots)")


@pytest.mark.parametrize("smoother", ["block211", "rbgs"])
def test_poisson3d_129_against_the_oracle(native, smoother):
    """129^3 is the smallest grid on which the LDS-sharing tile kernels and the fused residual + norm pass of the solver
    loop run by default: whole solves against the oracle (which restates the fused norm's summation order): same
    iteration count, residual history to round-off, solution bit-identical."""
    from oracle import oracle
    lo, hi = 5, 7
    pd = problems.poisson_3d(lo, hi)
    if smoother == "rbgs":
        sm = lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.15)
    else:
        sm = lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 1, 1), 0.8)
    ops = vcycle(lo, hi, 2, 2, sm)
    hg, ho = native.create(pd), oracle.create(pd)
    a, b = hg.evaluate(ops, 1, 0, 128), ho.evaluate(ops, 1, 0, 128)
    assert a.status == b.status == abi.SOLVE_CONVERGED and a.n_iterations == b.n_iterations
    assert np.max(np.abs(a.history - b.history) / b.history) <= 1e-12
    assert np.array_equal(hg.get_field(hi, abi.F_SOL), ho.get_field(hi, abi.F_SOL))
    assert np.array_equal(hg.get_field(hi, abi.F_RES), ho.get_field(hi, abi.F_RES))
    hg.close()
    ho.close()
