"""Every peephole of the op-program interpreter (csrc/impl_peephole.inl), systematically: for each one the MINIMAL op sequence
that triggers it, with each kind of smoother it can meet (point Jacobi, red-black, a two-cell block) and under BOTH `with jacobi`
semantics — the class of bug ADVICE r2 #1 was: a fusion that swallows an op whose sweep the reference pipeline's semantics
discards — compared three ways: default build, the same run with the peephole's switch off (EVOMG_NO_*), and the CPU oracle.
Non-cyclic form: `run_ops` on random fields, every field of every level bit for bit. Cyclic form (fusions that need "the program is
the body of the solver loop"): whole solves, identical histories between the two GPU runs, <= 1e-10 against the oracle, the
returned iterate bit for bit."""
import dataclasses

import numpy as np
import pytest

from evostencils_amd import abi, problems

pytestmark = pytest.mark.gpu
SEMANTICS = ["textbook", "exastencils-v1.1"]
SMOOTHERS = {
    "jacobi": lambda L, om=0.8: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1, 1, 1), om),
    "rbgs": lambda L, om=1.15: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), om),
    "block": lambda L, om=0.9: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 1, 1), om),
}


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()
    return nat


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as orc
    return orc


def residual(L):
    return abi.make_op(abi.OP_RESIDUAL, L, abi.F_RES, abi.F_SOL, abi.F_RHS)


def restrict(L):
    return abi.make_op(abi.OP_RESTRICT, L, abi.F_RHS, abi.F_RES)


def prolong(L, om=0.95):
    return abi.make_op(abi.OP_PROLONG_ADD, L, abi.F_SOL, abi.F_SOL, omega=om)


def zero(L, f=abi.F_SOL):
    return abi.make_op(abi.OP_SET_ZERO, L, f)


def coarse(lo):
    return [abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS), zero(lo, abi.F_CORR), abi.make_op(abi.OP_COARSE_SOLVE, lo)]


# name -> (dimension, lo, hi, switch that disables the peephole, ops(smoother, lo, hi) -> program); the level the pattern sits
# on is chosen so that the kernel family behind the peephole serves it (forced thresholds in `run` below)
def _top(lo, hi):
    return hi


CLASSICAL_CG = {"cgs_kind": abi.CGS_CG_CLASSICAL}      # optional sixth entry of a case: fields of the problem descriptor

NON_CYCLIC = {
    "residual_restrict_2d": (2, 4, 7, "EVOMG_NO_FUSE", lambda s, lo, hi: [residual(hi), restrict(hi - 1), residual(hi - 1), restrict(hi - 2)]),
    "prolong_smooth_2d": (2, 4, 7, "EVOMG_NO_FUSE", lambda s, lo, hi: [prolong(hi), s(hi), prolong(hi - 1), s(hi - 1), s(hi - 1)]),
    "zero_elision_2d": (2, 4, 7, "EVOMG_NO_ZERO_ELISION", lambda s, lo, hi: [zero(hi - 1), s(hi - 1), zero(hi - 2), s(hi - 2), s(hi - 2), residual(hi - 2)]),
    "copy_zero_coarse_solve_2d": (2, 4, 7, "EVOMG_NO_FUSE", lambda s, lo, hi: [s(lo + 1), residual(lo + 1), restrict(lo)] + coarse(lo) + [prolong(lo + 1), s(lo + 1)]),
    # the classical register CG (cgs_kind CGS_CG_CLASSICAL) and its own switch; the single-reduction form (the default kind)
    # folds wherever anything folds: EVOMG_NO_FUSE
    "copy_zero_register_cg_65_2d": (2, 6, 8, "EVOMG_NO_CG_FOLD", lambda s, lo, hi: [s(lo + 1), residual(lo + 1), restrict(lo)] + coarse(lo) + [prolong(lo + 1), s(lo + 1)], CLASSICAL_CG),
    "copy_zero_register_cg_129_2d": (2, 7, 8, "EVOMG_NO_CG_FOLD", lambda s, lo, hi: [s(lo + 1), residual(lo + 1), restrict(lo)] + coarse(lo) + [prolong(lo + 1), s(lo + 1)], CLASSICAL_CG),
    "copy_zero_one_wave_cg_2d": (2, 4, 7, "EVOMG_NO_FUSE", lambda s, lo, hi: [s(lo + 1), residual(lo + 1), restrict(lo)] + coarse(lo) + [prolong(lo + 1), s(lo + 1)], CLASSICAL_CG),
    "copy_zero_single_reduction_cg_65_2d": (2, 6, 8, "EVOMG_NO_FUSE", lambda s, lo, hi: [s(lo + 1), residual(lo + 1), restrict(lo)] + coarse(lo) + [prolong(lo + 1), s(lo + 1)]),
    "copy_zero_single_reduction_cg_129_2d": (2, 7, 8, "EVOMG_NO_FUSE", lambda s, lo, hi: [s(lo + 1), residual(lo + 1), restrict(lo)] + coarse(lo) + [prolong(lo + 1), s(lo + 1)]),
    "lds_tiles_2d": (2, 3, 6, "EVOMG_NO_SMALL2D", lambda s, lo, hi: [s(hi), s(hi - 1), prolong(hi), s(hi), zero(hi - 2), s(hi - 2), s(hi - 2)]),
    "strip_block_2d": (2, 4, 7, "EVOMG_NO_STRIP_BLOCK", lambda s, lo, hi: [s(hi), s(hi), residual(hi), s(hi - 1)]),
    "block_fast_2d": (2, 4, 7, "EVOMG_NO_BLOCK_FAST", lambda s, lo, hi: [s(hi), s(hi), residual(hi), s(hi - 1)]),
    "boundary_mirror_2d": (2, 4, 7, "EVOMG_NO_BOUNDARY_MIRROR", lambda s, lo, hi: [s(hi), s(hi), s(hi), prolong(hi), s(hi)]),
    "double_sweep_3d": (3, 2, 5, "EVOMG_NO_TILE3D_TWICE", lambda s, lo, hi: [s(hi), s(hi), residual(hi), s(hi - 1), s(hi - 1), s(hi - 1)]),
    "zero_double_sweep_3d": (3, 2, 5, "EVOMG_NO_TWICE_XZERO", lambda s, lo, hi: [zero(hi - 1), s(hi - 1), s(hi - 1), residual(hi - 1), s(hi - 1), zero(hi - 2), s(hi - 2), s(hi - 2),
                                                                                  s(hi - 2), residual(hi - 2)]),
    "prolong_double_sweep_3d": (3, 2, 5, "EVOMG_NO_TILE3D_PRO", lambda s, lo, hi: [prolong(hi), s(hi), s(hi), prolong(hi - 1), s(hi - 1), s(hi - 1)]),
    "small_level_chain_3d": (3, 2, 4, "EVOMG_NO_SMALL3D", lambda s, lo, hi: [zero(hi), s(hi), s(hi), residual(hi), restrict(hi - 1), prolong(hi), s(hi)]),
    "tiles_3d": (3, 2, 5, "EVOMG_NO_TILE3D", lambda s, lo, hi: [s(hi), residual(hi), restrict(hi - 1), s(hi - 1), prolong(hi), s(hi)]),
    "boundary_mirror_3d": (3, 2, 5, "EVOMG_NO_BOUNDARY_MIRROR", lambda s, lo, hi: [s(hi), s(hi), s(hi), prolong(hi), s(hi)]),
}


def v_cycle(s, lo, hi, pre, post):
    ops = []
    for L in range(hi, lo, -1):
        ops += [s(L)] * pre + [residual(L), restrict(L - 1)]
        if L - 1 > lo:
            ops.append(zero(L - 1))
    ops += coarse(lo)
    for L in range(lo + 1, hi + 1):
        ops += [prolong(L, 1.0)] + [s(L)] * post
    return ops


CYCLIC = {
    "sweep_residual_restrict_2d": (2, 4, 7, "EVOMG_NO_TAIL_RESTRICT", lambda s, lo, hi: v_cycle(s, lo, hi, 1, 1)),
    "two_sweeps_residual_restrict_2d": (2, 3, 6, "EVOMG_NO_SMALL2D", lambda s, lo, hi: v_cycle(s, lo, hi, 2, 1)),
    # round 6: on strip levels two red-black sweeps that start from zero + residual + restriction in ONE launch, the first sweep's
    # result formed from f (evomg_strip2d<.., XGEN>): levels 8 and 7 of this cycle (257^2, 129^2; 65^2 is an LDS tile)
    "two_sweeps_from_zero_on_strips_2d": (2, 5, 9, "EVOMG_NO_XGEN", lambda s, lo, hi: v_cycle(s, lo, hi, 2, 1)),
    "tail_norm_2d": (2, 4, 7, "EVOMG_NO_TAIL_NORM", lambda s, lo, hi: v_cycle(s, lo, hi, 1, 2)),
    "dead_residual_2d": (2, 4, 7, "EVOMG_NO_FUSE", lambda s, lo, hi: v_cycle(s, lo, hi, 0, 1)),
    "speculation_2d": (2, 4, 7, "EVOMG_NO_SPECULATION", lambda s, lo, hi: v_cycle(s, lo, hi, 2, 1)),
    "register_cg_fold_cycle_2d": (2, 6, 8, "EVOMG_NO_CG_FOLD", lambda s, lo, hi: v_cycle(s, lo, hi, 1, 1), CLASSICAL_CG),
    "single_reduction_cg_fold_cycle_2d": (2, 7, 9, "EVOMG_NO_FUSE", lambda s, lo, hi: v_cycle(s, lo, hi, 1, 1)),
    "cross_iteration_2d": (2, 4, 8, "EVOMG_NO_XFUSE", lambda s, lo, hi: v_cycle(s, lo, hi, 1, 1)),
    "cross_iteration_plain_2d": (2, 4, 8, "EVOMG_NO_XFUSE", lambda s, lo, hi: v_cycle(s, lo, hi, 2, 2)),
    "residual_restrict_3d": (3, 2, 5, "EVOMG_NO_TILE3D_RR", lambda s, lo, hi: v_cycle(s, lo, hi, 1, 1)),
    "double_sweep_cycle_3d": (3, 2, 5, "EVOMG_NO_TILE3D_TWICE", lambda s, lo, hi: v_cycle(s, lo, hi, 2, 2)),
    "residual_norm_3d": (3, 2, 5, "EVOMG_NO_FUSE", lambda s, lo, hi: v_cycle(s, lo, hi, 2, 1)),
    "speculation_3d": (3, 2, 5, "EVOMG_NO_SPECULATION", lambda s, lo, hi: v_cycle(s, lo, hi, 1, 1)),
}
FORCED = {"EVOMG_STRIP_MIN_N": "17", "EVOMG_STRIP3D_MIN_N": "9", "EVOMG_XFUSE_MIN_N": "65", "EVOMG_STRIP_ROWS": "16", "EVOMG_STRIP_PLANES": "8"}
KNOBS = sorted({v[3] for v in list(NON_CYCLIC.values()) + list(CYCLIC.values())})


def problem(dim, lo, hi, semantics, **kw):
    pd = problems.poisson_2d(lo, hi) if dim == 2 else problems.poisson_3d(lo, hi)
    return dataclasses.replace(pd, jacobi_semantics=abi.JACOBI_SEMANTICS[semantics], **kw)


def environment(monkeypatch, knob, fused):
    """`knob`: EVOMG_NO_X switches a default-on peephole off; +EVOMG_X switches a default-off one on."""
    for k in KNOBS:
        monkeypatch.delenv(k.lstrip("+"), raising=False)
    for k, v in FORCED.items():
        monkeypatch.setenv(k, v)
    if knob.startswith("+"):
        if fused:
            monkeypatch.setenv(knob[1:], "1")
    elif not fused:
        monkeypatch.setenv(knob, "1")


@pytest.mark.parametrize("semantics", SEMANTICS)
@pytest.mark.parametrize("kind", sorted(SMOOTHERS))
@pytest.mark.parametrize("name", sorted(NON_CYCLIC))
def test_peephole_on_random_fields(native, oracle, monkeypatch, name, kind, semantics):
    dim, lo, hi, knob, build = NON_CYCLIC[name][:5]
    ops = build(SMOOTHERS[kind], lo, hi)
    pd = problem(dim, lo, hi, semantics, **(NON_CYCLIC[name][5] if len(NON_CYCLIC[name]) > 5 else {}))
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{name}/{kind}".encode()))
    state = {}
    results = []
    for which in ("fused", "switched_off", "oracle"):
        environment(monkeypatch, knob, which != "switched_off")
        h = (oracle if which == "oracle" else native).create(pd)
        for L in range(lo, hi + 1):
            for f in (abi.F_SOL, abi.F_RHS, abi.F_RES):
                if (L, f) not in state:
                    a = rng.standard_normal(h.field_shape(L))
                    if f == abi.F_SOL and L < hi:
                        # error fields have a zero boundary in every cycle (what the zero-input variants rely on)
                        edge = np.ones(a.shape, bool)
                        edge[(slice(1, -1),) * dim] = False
                        a[edge] = 0.0
                    state[(L, f)] = a
                h.set_field(L, f, state[(L, f)])
        h.run_ops(ops)
        results.append({(L, f): h.get_field(L, f) for L in range(lo, hi + 1) for f in (abi.F_SOL, abi.F_RHS, abi.F_RES)})
        h.close()
    fused, plain, want = results
    for key in want:
        assert np.array_equal(fused[key], want[key], equal_nan=True), ("fused vs oracle", name, kind, semantics, key)
        assert np.array_equal(plain[key], want[key], equal_nan=True), ("switched off vs oracle", name, kind, semantics, key)


@pytest.mark.parametrize("semantics", SEMANTICS)
@pytest.mark.parametrize("kind", sorted(SMOOTHERS))
@pytest.mark.parametrize("name", sorted(CYCLIC))
def test_peephole_in_the_solver_loop(native, oracle, monkeypatch, name, kind, semantics):
    dim, lo, hi, knob, build = CYCLIC[name][:5]
    if dim == 3 and kind == "block" and hi > 4:
        hi = 5
    ops = build(SMOOTHERS[kind], lo, hi)
    pd = problem(dim, lo, hi, semantics, max_iterations=5, **(CYCLIC[name][5] if len(CYCLIC[name]) > 5 else {}))
    outs = []
    for which in ("fused", "switched_off", "oracle"):
        environment(monkeypatch, knob, which != "switched_off")
        h = (oracle if which == "oracle" else native).create(pd)
        out = h.evaluate(ops, 2 if which != "oracle" else 1, 0, 16)
        outs.append((out, h.get_field(hi, abi.F_SOL), h.get_field(hi, abi.F_RES)))
        h.close()
    (a, ua, ra), (b, ub, rb), (c, uc, rc) = outs
    assert a.status == b.status == c.status and a.n_iterations == b.n_iterations == c.n_iterations
    assert np.array_equal(a.history, b.history, equal_nan=True)
    finite = np.isfinite(c.history) & (c.history > 0)
    assert np.array_equal(np.isfinite(a.history), np.isfinite(c.history))
    assert np.max(np.abs(a.history[finite] - c.history[finite]) / c.history[finite]) <= 1e-10
    assert np.array_equal(ua, ub, equal_nan=True) and np.array_equal(ua, uc, equal_nan=True)
    assert np.array_equal(ra, rb, equal_nan=True) and np.array_equal(ra, rc, equal_nan=True)


@pytest.mark.parametrize("level", [3, 4, 5, 6])
def test_one_launch_red_black_sweep_on_small_3d_levels(native, oracle, monkeypatch, level):
    """3D levels below the tile threshold (9^3 .. 65^3, DEFAULT thresholds — the forced ones above hand these sizes to the
    tiles): both colours of a red-black sweep in one launch (evomg_rbgs3d_block: blocks of 16 x 8 x 8 with recomputed halo
    reds; 63 and 31 inner nodes are not multiples of the block edges) vs two thread-per-node launches vs the oracle, on random
    fields with a non-zero boundary, three sweeps with different weights, a residual behind them."""
    for k in list(FORCED) + KNOBS:
        monkeypatch.delenv(k.lstrip("+"), raising=False)
    pd = problem(3, level - 1, level, "textbook")
    rb = lambda om: abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), om)
    ops = [rb(1.0), rb(1.25), rb(0.7), residual(level)]
    rng = np.random.default_rng(level)
    fields = None
    got = []
    for which in ("fused", "switched_off", "oracle"):
        if which == "switched_off":
            monkeypatch.setenv("EVOMG_NO_RBGS3D_BLOCK", "1")
        h = (oracle if which == "oracle" else native).create(pd)
        if fields is None:
            fields = {f: rng.standard_normal(h.field_shape(level)) for f in (abi.F_SOL, abi.F_RHS)}
        for f, a in fields.items():
            h.set_field(level, f, a)
        if which == "fused":
            assert h.kernel_name(ops[0]) == "evomg_rbgs3d_block"
        h.run_ops(ops)
        got.append((h.get_field(level, abi.F_SOL), h.get_field(level, abi.F_RES)))
        h.close()
    for u, r in got[:2]:
        assert np.array_equal(u, got[2][0]) and np.array_equal(r, got[2][1])


@pytest.mark.parametrize("level", [5, 6])
def test_zero_and_prolongation_ride_on_the_small_3d_sweep(native, oracle, monkeypatch, level):
    """SET_ZERO + sweep and PROLONG_ADD + sweep on 33^3 / 65^3 at default thresholds: the zero is elided and the correction is
    formed on the one-launch sweep's load (evomg_rbgs3d_block: zero_first / coarse_in) — against the same ops as separate
    launches (EVOMG_NO_FUSE) and against the oracle, on fields with a zero boundary (what an error field of a cycle has)."""
    for k in list(FORCED) + KNOBS:
        monkeypatch.delenv(k.lstrip("+"), raising=False)
    pd = problem(3, level - 1, level, "textbook")
    rb = lambda om: abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), om)
    ops = [zero(level), rb(1.25), rb(1.25), residual(level), prolong(level, 0.9), rb(1.0), residual(level)]
    rng = np.random.default_rng(100 + level)
    fields = {}
    got = []
    for which in ("fused", "switched_off", "oracle"):
        if which == "switched_off":
            monkeypatch.setenv("EVOMG_NO_FUSE", "1")
        h = (oracle if which == "oracle" else native).create(pd)
        for L in (level - 1, level):
            for f in (abi.F_SOL, abi.F_RHS):
                if (L, f) not in fields:
                    a = rng.standard_normal(h.field_shape(L))
                    if f == abi.F_SOL:
                        edge = np.ones(a.shape, bool)
                        edge[(slice(1, -1),) * 3] = False
                        a[edge] = 0.0
                    fields[(L, f)] = a
                h.set_field(L, f, fields[(L, f)])
        h.run_ops(ops)
        got.append((h.get_field(level, abi.F_SOL), h.get_field(level, abi.F_RES)))
        h.close()
    for u, r in got[:2]:
        assert np.array_equal(u, got[2][0]) and np.array_equal(r, got[2][1])


@pytest.mark.parametrize("knob", ["EVOMG_NO_RBGS3D_BLOCK", "EVOMG_NO_RR3D_BLOCK"])
@pytest.mark.parametrize("kind", ["rbgs", "jacobi"])
def test_small_3d_levels_in_the_solver_loop_at_default_thresholds(native, oracle, monkeypatch, kind, knob):
    """The reference's 3D example (levels 2..6, 65^3) with nothing forced: the one-launch red-black sweeps (with the zero and the
    prolongation riding on them) and the one-launch residual + restriction of the levels below the tile threshold, as the body
    of the solver loop — with each of them switched off, and against the oracle."""
    for k in list(FORCED) + KNOBS:
        monkeypatch.delenv(k.lstrip("+"), raising=False)
    lo, hi = 2, 6
    ops = v_cycle(SMOOTHERS[kind], lo, hi, 2, 1)
    pd = problem(3, lo, hi, "textbook", max_iterations=4)
    outs = []
    for which in ("fused", "switched_off", "oracle"):
        if which == "switched_off":
            monkeypatch.setenv(knob, "1")
        h = (oracle if which == "oracle" else native).create(pd)
        out = h.evaluate(ops, 1, 0, 16)
        outs.append((out, h.get_field(hi, abi.F_SOL), h.get_field(hi, abi.F_RES)))
        h.close()
    (a, ua, ra), (b, ub, rb), (c, uc, rc) = outs
    assert a.n_iterations == b.n_iterations == c.n_iterations
    assert np.array_equal(a.history, b.history) and np.array_equal(a.history, c.history)
    assert np.array_equal(ua, ub) and np.array_equal(ua, uc) and np.array_equal(ra, rb) and np.array_equal(ra, rc)
