"""SMOOTH, SMOOTH, RESIDUAL, RESTRICT on a strip level whose sweeps start from the zero field — the pre-smoothing of a V(2,.)
cycle on every level below the finest, and on the finest one inside a preconditioner — as ONE launch: the rows the first sweep
would leave in memory are formed from f inside the second sweep's launch (csrc/evomg_strip.h: XGEN), Robin columns included.
Against the same solves with EVOMG_NO_XGEN (two launches), bit for bit, on the reference's Helmholtz cycle
(Helmholtz/2D_FD_Helmholtz_fromL3.exa3:202-272: V(2,1) red-black w = 0.6 inside BiCGStab) at sizes where complex levels run on
strips (513^2 and 1025^2), on real V(2,1) / V(2,2) cycles with forced chunk heights, and against the oracle where that is
affordable."""
import dataclasses

import numpy as np
import pytest

from evostencils_amd import abi, candidates, problems

pytestmark = pytest.mark.gpu
REL_TOL = 1e-10


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()
    return nat


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as orc
    return orc


def _solve(native, pd, ops, hi, monkeypatch, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    h = native.create(pd)
    out = h.evaluate(ops, 1, 0, 256)
    fields = (h.get_field(hi, abi.F_SOL), h.get_field(hi, abi.F_CORR), h.get_field(hi, abi.F_RES))
    h.close()
    for k in env:
        monkeypatch.delenv(k)
    return out, fields


# (3, 7) and (4, 8): no strip level at all — the levels of several LDS tiles (65^2 .. 257^2 complex) run both sweeps, the residual
# and the restriction in each tile's launch with a halo of six nodes (csrc/evomg_small2d.h), same switch
@pytest.mark.parametrize("lo,hi,cap", [(5, 9, 6), (6, 10, 3), (7, 10, 4), (3, 7, 40), (4, 8, 12)])
def test_helmholtz_preconditioner_with_the_first_sweep_formed_from_f(native, oracle, monkeypatch, lo, hi, cap):
    pd = dataclasses.replace(problems.helmholtz_2d(lo, hi, k=0.625 * (1 << hi)), max_iterations=cap)
    ops = candidates.v_cycle_ops(lo, hi, 2, 1, omega=0.6)
    a, fa = _solve(native, pd, ops, hi, monkeypatch, {})
    b, fb = _solve(native, pd, ops, hi, monkeypatch, {"EVOMG_NO_XGEN": "1"})
    assert a.status == b.status and a.n_iterations == b.n_iterations == cap
    assert np.array_equal(a.history, b.history)
    for x, y in zip(fa, fb):
        assert np.array_equal(x, y)
    if hi <= 9:
        ho = oracle.create(pd)
        o = ho.evaluate(ops, 1, 0, 256)
        assert o.n_iterations == a.n_iterations
        assert (np.abs(a.history - o.history) / o.history).max() <= REL_TOL
        so = ho.get_field(hi, abi.F_CORR)
        assert np.abs(fa[1] - so).max() <= REL_TOL * np.abs(so).max()


@pytest.mark.parametrize("rows", ["0", "8", "22"])
@pytest.mark.parametrize("pre,post,omega", [(2, 1, 1.0), (2, 2, 0.8)])
def test_real_cycles_with_two_pre_sweeps(native, oracle, monkeypatch, pre, post, omega, rows):
    lo, hi = 5, 9
    pd = dataclasses.replace(problems.poisson_2d(lo, hi), max_iterations=4)
    ops = candidates.v_cycle_ops(lo, hi, pre, post, omega=omega)
    env = {} if rows == "0" else {"EVOMG_STRIP_ROWS": rows}
    a, fa = _solve(native, pd, ops, hi, monkeypatch, env)
    b, fb = _solve(native, pd, ops, hi, monkeypatch, dict(env, EVOMG_NO_XGEN="1"))
    assert a.status == b.status and a.n_iterations == b.n_iterations
    assert np.array_equal(a.history, b.history)
    assert np.array_equal(fa[0], fb[0]) and np.array_equal(fa[2], fb[2])
    if rows == "0":
        ho = oracle.create(pd)
        o = ho.evaluate(ops, 1, 0, 256)
        assert o.n_iterations == a.n_iterations
        assert (np.abs(a.history - o.history) / o.history).max() <= REL_TOL
        assert np.array_equal(fa[0], ho.get_field(hi, abi.F_SOL))
