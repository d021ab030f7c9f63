"""Cross-iteration fusion on the finest 2D level (csrc/evomg_strip_cross.h): the cycle's last red-black sweep (+ prolongation,
+ the loop's norm) and the next cycle's first sweep (+ residual + restriction) as ONE speculative launch. It must not change
a bit: residual histories, the returned iterate, the residual field and the coarse right-hand side are compared with the same
solve under EVOMG_NO_XFUSE, and with the oracle where it finishes in seconds — for every shape of the plan (with / without the
folded prolongation, with / without the restriction tail), different relaxation factors on the two sweeps, chunk heights of
16 .. 256 rows, solves that converge, solves that run into the iteration cap and a cap of one."""
import dataclasses

import numpy as np
import pytest

from evostencils_amd import abi, problems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()
    return nat


def rb(L, om):
    return abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), om)


def jac(L, om):
    return abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1, 1, 1), om)


def cycle(lo, hi, pre, post, cgc=1.0):
    """pre / post: per level (finest first) lists of smoother factories."""
    ops = []
    for k, L in enumerate(range(hi, lo, -1)):
        ops += [s(L) for s in pre[min(k, len(pre) - 1)]]
        ops.append(abi.make_op(abi.OP_RESIDUAL, L, abi.F_RES, abi.F_SOL, abi.F_RHS))
        ops.append(abi.make_op(abi.OP_RESTRICT, L - 1, abi.F_RHS, abi.F_RES))
        if L - 1 > lo:
            ops.append(abi.make_op(abi.OP_SET_ZERO, L - 1, abi.F_SOL))
    ops.append(abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS))
    ops.append(abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR))
    ops.append(abi.make_op(abi.OP_COARSE_SOLVE, lo))
    for L in range(lo + 1, hi + 1):
        k = hi - L
        ops.append(abi.make_op(abi.OP_PROLONG_ADD, L, abi.F_SOL, abi.F_SOL, omega=cgc))
        ops += [s(L) for s in post[min(k, len(post) - 1)]]
    return ops


R = lambda om: (lambda L: rb(L, om))
J = lambda om: (lambda L: jac(L, om))
PLANS = {
    "v11_pro_tailr": ([[R(1.0)]], [[R(1.0)]]),
    "v11_two_omegas": ([[R(1.15)]], [[R(0.9)]]),
    "v22_plain": ([[R(1.0), R(1.1)]], [[R(0.95), R(1.05)]]),            # no folded prolongation, no restriction tail
    "v12_tailr_only": ([[R(1.2)]], [[R(1.0), R(0.8)]]),
    "v21_pro_only": ([[R(1.0), R(1.0)]], [[R(1.1)]]),
    "jacobi_between": ([[R(1.0)], [J(0.8)]], [[R(1.0)], [J(0.8), R(1.0)]]),   # coarser levels differ; finest still fuses
    # every weight the launch sees is 1 (BASELINE's C2 cycle): the kernel's unit-weight form (round 6)
    "v11_unit": ([[R(1.0)]], [[R(1.0)]], 1.0),
    "v22_unit_plain": ([[R(1.0), R(1.0)]], [[R(1.0), R(1.0)]], 1.0),
    "v12_unit_tailr_only": ([[R(1.0)]], [[R(0.9), R(1.0)]], 1.0),
    "v21_unit_pro_only": ([[R(1.0), R(1.1)]], [[R(1.0)]], 1.0),
}


def solve(native, pd, ops, fields, env, monkeypatch):
    for k in ("EVOMG_NO_XFUSE", "EVOMG_XFUSE_MIN_N", "EVOMG_STRIP_ROWS", "EVOMG_NO_CROSS_UNIT", "EVOMG_NO_NORM_FOLD"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    h = native.create(pd)
    out = h.evaluate(ops, 2, 0, 128)
    got = [h.get_field(L, f) for L, f in fields]
    h.close()
    return out, got


@pytest.mark.parametrize("plan", sorted(PLANS))
@pytest.mark.parametrize("lo,hi,rows", [(5, 8, "0"), (4, 9, "0"), (6, 9, "32"), (6, 10, "0")])
def test_cross_fusion_does_not_change_a_bit(native, monkeypatch, plan, lo, hi, rows):
    pre, post, cgc = (PLANS[plan] + (0.95,))[:3]
    pd = problems.poisson_2d(lo, hi)
    ops = cycle(lo, hi, pre, post, cgc=cgc)
    fields = [(hi, abi.F_SOL), (hi, abi.F_RES), (hi - 1, abi.F_RHS), (hi - 1, abi.F_SOL)]
    env = {"EVOMG_XFUSE_MIN_N": "65"}
    if rows != "0":
        env["EVOMG_STRIP_ROWS"] = rows
    a, fa = solve(native, pd, ops, fields, env, monkeypatch)
    b, fb = solve(native, pd, ops, fields, {**env, "EVOMG_NO_XFUSE": "1"}, monkeypatch)
    assert a.status == b.status and a.n_iterations == b.n_iterations and a.n_iterations >= 2
    assert np.array_equal(a.history, b.history)
    for x, y, (L, f) in zip(fa, fb, fields):
        assert np.array_equal(x, y), (L, f)
    if "unit" in plan:
        # the unit-weight form against the general power-of-two form of the same launch
        c, fc = solve(native, pd, ops, fields, {**env, "EVOMG_NO_CROSS_UNIT": "1"}, monkeypatch)
        assert np.array_equal(a.history, c.history)
        for x, y, (L, f) in zip(fa, fc, fields):
            assert np.array_equal(x, y), (L, f)


@pytest.mark.parametrize("rows", ["16", "48", "80", "96", "256"])     # 256: the height the launch takes when the device is full
@pytest.mark.parametrize("cap", [1, 2, 5])
def test_cross_fusion_at_the_iteration_cap_and_odd_chunk_heights(native, monkeypatch, rows, cap):
    lo, hi = 5, 9
    pd = dataclasses.replace(problems.poisson_2d(lo, hi), max_iterations=cap)
    ops = cycle(lo, hi, [[R(1.0)]], [[R(1.0)]])
    fields = [(hi, abi.F_SOL), (hi, abi.F_RES), (hi - 1, abi.F_RHS)]
    env = {"EVOMG_XFUSE_MIN_N": "65", "EVOMG_STRIP_ROWS": rows}
    a, fa = solve(native, pd, ops, fields, env, monkeypatch)
    b, fb = solve(native, pd, ops, fields, {**env, "EVOMG_NO_XFUSE": "1"}, monkeypatch)
    assert a.status == b.status == abi.SOLVE_MAXITER and a.n_iterations == b.n_iterations == cap
    assert np.array_equal(a.history, b.history)
    for x, y in zip(fa, fb):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("plan", ["v11_pro_tailr", "v22_plain", "v12_tailr_only", "v22_unit_plain", "v21_unit_pro_only"])
def test_cross_fusion_against_the_oracle(native, monkeypatch, plan):
    """1025^2, five levels (the fusion's default threshold is 2049^2, where the oracle takes a minute per solve; the C2 / C5
    tests at full size run through it by default): same iterations, histories to 1e-10, the returned iterate bit for bit."""
    from oracle import oracle
    lo, hi = 6, 10
    pre, post = PLANS[plan][:2]
    pd = problems.poisson_2d(lo, hi)
    ops = cycle(lo, hi, pre, post)                      # (correction weight 1: "v11_pro_tailr" is the all-unit C2 cycle here)
    a, (ua,) = solve(native, pd, ops, [(hi, abi.F_SOL)], {"EVOMG_XFUSE_MIN_N": "1025"}, monkeypatch)
    ho = oracle.create(pd)
    b = ho.evaluate(ops, 1, 0, 128)
    assert a.status == b.status == abi.SOLVE_CONVERGED and a.n_iterations == b.n_iterations
    assert np.all(np.abs(a.history - b.history) <= 1e-10 * b.history)
    assert np.array_equal(ua, ho.get_field(hi, abi.F_SOL))
    ho.close()


@pytest.mark.parametrize("lo,hi", [(5, 8), (5, 9), (6, 10), (7, 11)])
def test_the_norm_in_one_launch_is_the_norm_in_two(native, monkeypatch, lo, hi):
    """Round 6: the cross launch's partial sums and their sum in ONE launch (the last workgroup to arrive adds them up in
    evomg_dot_final's order) against the two launches it replaces — and, sixty solves on one handle, the arrival ticket back at
    zero every time."""
    pd = problems.poisson_2d(lo, hi)
    ops = cycle(lo, hi, [[R(1.0)]], [[R(1.1)]], cgc=0.9)
    fields = [(hi, abi.F_SOL), (hi - 1, abi.F_RHS)]
    env = {"EVOMG_XFUSE_MIN_N": "65"}
    a, fa = solve(native, pd, ops, fields, env, monkeypatch)
    b, fb = solve(native, pd, ops, fields, {**env, "EVOMG_NO_NORM_FOLD": "1"}, monkeypatch)
    assert a.status == b.status and a.n_iterations == b.n_iterations >= 2
    assert np.array_equal(a.history, b.history)
    for x, y in zip(fa, fb):
        assert np.array_equal(x, y)
    if hi <= 9:
        monkeypatch.delenv("EVOMG_NO_NORM_FOLD", raising=False)
        monkeypatch.setenv("EVOMG_XFUSE_MIN_N", "65")
        h = native.create(pd)
        for _ in range(20):
            out = h.evaluate(ops, 3, 0, 128)
            assert np.array_equal(out.history, a.history)
        h.close()


def test_the_plan_is_taken_only_for_red_black_ends(native, monkeypatch):
    """Cycles whose first or last finest-level smoother is not a red-black point sweep run exactly as before (the kernel name of
    the profile does not matter here: results with and without the switch are trivially the same path); and a wall limit that
    fires in the first iteration still returns that iteration's iterate."""
    lo, hi = 5, 9
    pd = problems.poisson_2d(lo, hi)
    for pre, post in (([[J(0.8)]], [[R(1.0)]]), ([[R(1.0)]], [[J(0.8)]])):
        ops = cycle(lo, hi, pre, post)
        a, fa = solve(native, pd, ops, [(hi, abi.F_SOL)], {"EVOMG_XFUSE_MIN_N": "65"}, monkeypatch)
        b, fb = solve(native, pd, ops, [(hi, abi.F_SOL)], {"EVOMG_XFUSE_MIN_N": "65", "EVOMG_NO_XFUSE": "1"}, monkeypatch)
        assert np.array_equal(a.history, b.history) and np.array_equal(fa[0], fb[0])
    ops = cycle(lo, hi, [[R(1.0)]], [[R(1.0)]])
    monkeypatch.setenv("EVOMG_XFUSE_MIN_N", "65")
    h = native.create(pd)
    out = h.evaluate(ops, 1, 1e-6, 128)                 # a wall limit of a nanosecond: stops after the first iteration
    assert out.status == abi.SOLVE_WALL_LIMIT and out.n_iterations == 1
    u = h.get_field(hi, abi.F_SOL)
    h.close()
    monkeypatch.setenv("EVOMG_NO_XFUSE", "1")
    h = native.create(dataclasses.replace(pd, max_iterations=1))
    ref = h.evaluate(ops, 1, 0, 128)
    assert np.array_equal(h.get_field(hi, abi.F_SOL), u) and ref.history[1] == out.history[1]
    h.close()
