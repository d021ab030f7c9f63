"""Randomised op programs, HIP backend vs CPU oracle through the C-ABI: sequences the hand-written tests do not spell out
(any order of smoothers, transfers, residuals, zeroing, copies and coarse solves on any level), on grids where the strip /
tile kernels and the interpreter's peephole fusions are active. Everything the ops write is compared BIT FOR BIT after
`run_ops`; as the body of the solver loop (cyclic: fusions with dead residual fields, tail norm, speculation) the residual
histories agree to 1e-10 and the solutions bit for bit."""
import numpy as np
import pytest

from evostencils_amd import abi, problems

pytestmark = pytest.mark.gpu

OMEGAS = [0.5, 0.6, 0.8, 0.9999999999999999, 1.0, 1.15]


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()
    return nat


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as orc
    return orc


def random_program(rng, dim, lo, hi, length):
    blocks2 = [(1, 1, 1), (2, 1, 1), (1, 2, 1), (2, 2, 1), (3, 1, 1)]
    blocks3 = [(1, 1, 1), (2, 1, 1), (1, 2, 1), (1, 1, 2), (2, 2, 1), (2, 2, 2)]
    ops = [abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS)]          # the coarse solver's right-hand side exists
    for _ in range(length):
        L = int(rng.integers(lo, hi + 1))
        kind = rng.choice(["smooth", "smooth", "smooth", "residual", "restrict", "prolong_add", "prolong_set", "zero", "copy", "coarse",
                           "smooth_res_restrict", "prolong_smooth", "smooth_twice", "prolong_smooth_twice"])
        om = float(rng.choice(OMEGAS))
        part = abi.PART_RED_BLACK if rng.random() < 0.5 else abi.PART_SINGLE
        block = (1, 1, 1) if part == abi.PART_RED_BLACK or rng.random() < 0.5 else (blocks2 if dim == 2 else blocks3)[int(rng.integers(0, 5))]
        smooth = abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, part, block, om)
        if kind == "smooth":
            ops.append(smooth)
        elif kind == "residual":
            ops.append(abi.make_op(abi.OP_RESIDUAL, L, abi.F_RES, abi.F_SOL, abi.F_RHS))
        elif kind == "restrict" and L < hi:
            ops.append(abi.make_op(abi.OP_RESTRICT, L, abi.F_RHS, abi.F_RES if rng.random() < 0.7 else abi.F_SOL))
        elif kind == "prolong_add" and L > lo:
            ops.append(abi.make_op(abi.OP_PROLONG_ADD, L, abi.F_SOL, abi.F_SOL, omega=om))
        elif kind == "prolong_set" and L > lo:
            ops.append(abi.make_op(abi.OP_PROLONG_SET, L, abi.F_RES, abi.F_SOL))
        elif kind == "zero" and L < hi:
            ops.append(abi.make_op(abi.OP_SET_ZERO, L, abi.F_SOL))
        elif kind == "copy":
            ops.append(abi.make_op(abi.OP_COPY, L, abi.F_RES, abi.F_RHS))
        elif kind == "coarse":
            ops += [abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS), abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR),
                    abi.make_op(abi.OP_COARSE_SOLVE, lo)]
        elif kind == "smooth_res_restrict" and L > lo:       # the shapes the peepholes look for
            ops += [smooth, abi.make_op(abi.OP_RESIDUAL, L, abi.F_RES, abi.F_SOL, abi.F_RHS), abi.make_op(abi.OP_RESTRICT, L - 1, abi.F_RHS, abi.F_RES)]
        elif kind == "prolong_smooth" and L > lo:
            ops += [abi.make_op(abi.OP_PROLONG_ADD, L, abi.F_SOL, abi.F_SOL, omega=om), smooth]
        elif kind == "smooth_twice":                          # temporal fusion of two identical sweeps (3D tiles)
            ops += [smooth, smooth]
        elif kind == "prolong_smooth_twice" and L > lo:
            ops += [abi.make_op(abi.OP_PROLONG_ADD, L, abi.F_SOL, abi.F_SOL, omega=om), smooth, smooth]
    return ops


def load_random(hg, ho, pd, rng):
    for L in range(pd.min_level, pd.max_level + 1):
        shape = hg.field_shape(L)
        for f in (abi.F_SOL, abi.F_RHS, abi.F_RES):
            a = rng.standard_normal(shape)
            hg.set_field(L, f, a)
            ho.set_field(L, f, a)


# (dim, lo, hi, seed): small grids with forced strips / tiles, and grids with several strips, y groups and z chunks
FUZZ = ([(2, 3, 7, s) for s in range(8)] + [(3, 2, 5, s) for s in range(8)] + [(2, 5, 9, s) for s in range(8, 12)]
        + [(3, 3, 6, s) for s in range(8, 11)] + [(3, 4, 7, 11), (2, 6, 10, 12)])


def with_semantics(pd, semantics):
    import dataclasses
    return dataclasses.replace(pd, jacobi_semantics=abi.JACOBI_SEMANTICS[semantics])


# the reference pipeline's `with jacobi` semantics (EVOMG_JACOBI_EXASTENCILS_V11: Single-partitioned sweeps are launched
# and their result discarded) is the drop-in class's default: every fusion has to leave such sweeps alone
FUZZ_V11 = [(d, lo, hi, s, "exastencils-v1.1") for d, lo, hi, s in FUZZ if s % 2 == 0 or s >= 8]


@pytest.mark.parametrize("dim,lo,hi,seed,semantics", [f + ("textbook",) for f in FUZZ] + FUZZ_V11)
def test_random_programs_are_bit_exact(native, oracle, dim, lo, hi, seed, semantics, monkeypatch):
    # strips / tiles and their fusions from 17^2 / 9^3 up, odd chunk sizes
    monkeypatch.setenv("EVOMG_STRIP_MIN_N", "17")
    monkeypatch.setenv("EVOMG_STRIP3D_MIN_N", "9")
    monkeypatch.setenv("EVOMG_STRIP_ROWS", str([0, 10, 16, 22][seed % 4]))
    monkeypatch.setenv("EVOMG_STRIP_PLANES", str([0, 4, 8, 12][seed % 4]))
    pd = with_semantics(problems.poisson_2d(lo, hi) if dim == 2 else problems.poisson_3d(lo, hi), semantics)
    rng = np.random.default_rng(1000 * dim + seed)
    hg, ho = native.create(pd), oracle.create(pd)
    ops = random_program(rng, dim, lo, hi, 24 if hi - lo <= 4 else 30)
    load_random(hg, ho, pd, rng)
    hg.run_ops(ops)
    ho.run_ops(ops)
    for L in range(lo, hi + 1):
        for f in (abi.F_SOL, abi.F_RHS, abi.F_RES):
            got, want = hg.get_field(L, f), ho.get_field(L, f)
            assert np.array_equal(got, want, equal_nan=True), (dim, seed, L, f, np.nanmax(np.abs(got - want)))
    hg.close()
    ho.close()


@pytest.mark.parametrize("lo,hi,seed,semantics", [(5, 7, 40, "textbook"), (5, 7, 41, "exastencils-v1.1"), (6, 8, 42, "textbook")])
def test_random_3d_programs_at_default_thresholds_are_bit_exact(native, oracle, lo, hi, seed, semantics):
    """Random op programs on 3D hierarchies that end on 129^3 / 257^3 with NOTHING forced: the marching tiles with their
    round-4 launch geometry (chunk depth by workgroup balance, strips of equal width), the double sweeps and the one-launch
    red-black sweeps of the levels below them as a candidate gets them by default (tools/fuzz_more.py 3dbig runs more seeds)."""
    pd = with_semantics(problems.poisson_3d(lo, hi), semantics)
    rng = np.random.default_rng(3000 + seed)
    hg, ho = native.create(pd), oracle.create(pd)
    ops = random_program(rng, 3, lo, hi, 28)
    load_random(hg, ho, pd, rng)
    hg.run_ops(ops)
    ho.run_ops(ops)
    for L in range(lo, hi + 1):
        for f in (abi.F_SOL, abi.F_RHS, abi.F_RES):
            got, want = hg.get_field(L, f), ho.get_field(L, f)
            assert np.array_equal(got, want, equal_nan=True), (seed, L, f, np.nanmax(np.abs(got - want)))
    hg.close()
    ho.close()


@pytest.mark.parametrize("lo,hi,seed,semantics", [(3, 7, s, "textbook") for s in range(4)] + [(4, 8, s, "exastencils-v1.1") for s in range(4, 7)])
def test_random_programs_on_helmholtz_are_bit_exact(native, oracle, lo, hi, seed, semantics):
    """The same random programs on the complex Helmholtz problem: Robin columns after every op that writes a field, the
    LDS-resident small-level sweeps, the one-wave BiCGStab on the coarsest grid — every field of every level bit for bit
    (tools/fuzz_more.py runs as many further seeds as one likes)."""
    pd = with_semantics(problems.helmholtz_2d(lo, hi, k=0.625 * (1 << hi)), semantics)
    rng = np.random.default_rng(500 + 31 * seed + lo)
    ops = random_program(rng, 2, lo, hi, 28)
    hg, ho = native.create(pd), oracle.create(pd)
    for L in range(lo, hi + 1):
        shape = hg.field_shape(L)
        for f in (abi.F_SOL, abi.F_RHS, abi.F_RES):
            a = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
            hg.set_field(L, f, a)
            ho.set_field(L, f, a)
    hg.run_ops(ops)
    ho.run_ops(ops)
    for L in range(lo, hi + 1):
        for f in (abi.F_SOL, abi.F_RHS, abi.F_RES):
            got, want = hg.get_field(L, f), ho.get_field(L, f)
            assert np.array_equal(got, want, equal_nan=True), (seed, L, f, np.nanmax(np.abs(got - want)))
    hg.close()
    ho.close()


CYCLES = ([(2, 3, 7, s) for s in range(6)] + [(3, 2, 5, s) for s in range(6)] + [(2, 5, 9, 6), (2, 5, 9, 7), (3, 3, 6, 8), (3, 4, 7, 9)])


@pytest.mark.parametrize("dim,lo,hi,seed,semantics", [c + ("textbook",) for c in CYCLES] + [c + ("exastencils-v1.1",) for c in CYCLES])
def test_random_cycles_in_the_solver_loop(native, oracle, dim, lo, hi, seed, semantics, monkeypatch):
    """A random V-shaped cycle (random smoothers and sweep counts per level, random weights) as the body of the solver
    loop: the cyclic-only fusions (dead residual fields, norm tails, speculation) against the oracle."""
    import dataclasses
    from test_full_size_gpu import vcycle
    monkeypatch.setenv("EVOMG_STRIP_MIN_N", "17")
    monkeypatch.setenv("EVOMG_STRIP3D_MIN_N", "9")
    monkeypatch.setenv("EVOMG_STRIP_PLANES", str([0, 4, 8][seed % 3]))
    rng = np.random.default_rng(77 * dim + seed)
    pd = dataclasses.replace(problems.poisson_2d(lo, hi) if dim == 2 else problems.poisson_3d(lo, hi), max_iterations=6,
                             jacobi_semantics=abi.JACOBI_SEMANTICS[semantics])
    blocks = [(1, 1, 1), (2, 1, 1), (1, 2, 1)]

    def smoother(L):
        part = abi.PART_RED_BLACK if rng.random() < 0.5 else abi.PART_SINGLE
        block = (1, 1, 1) if part == abi.PART_RED_BLACK else blocks[int(rng.integers(0, 3))]
        return abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, part, block, float(rng.choice(OMEGAS[:4])))

    per_level = {L: smoother(L) for L in range(lo, hi + 1)}
    ops = vcycle(lo, hi, int(rng.integers(0, 3)), int(rng.integers(1, 3)), lambda L: per_level[L])
    hg, ho = native.create(pd), oracle.create(pd)
    got, want = hg.evaluate(ops, 1, 0, 32), ho.evaluate(ops, 1, 0, 32)
    assert got.status == want.status and got.n_iterations == want.n_iterations
    finite = np.isfinite(want.history)                  # a cycle whose only smoothers are lost sweeps may diverge
    assert np.array_equal(finite, np.isfinite(got.history))
    assert np.max(np.abs(got.history[finite] - want.history[finite]) / want.history[finite]) <= 1e-10
    assert np.array_equal(hg.get_field(hi, abi.F_SOL), ho.get_field(hi, abi.F_SOL))
    hg.close()
    ho.close()


MORE = [(kind, lo, hi, seed) for seed in range(100, 120) for kind, lo, hi in (("poisson", 3, 7), ("helmholtz", 3, 7), ("helmholtz", 4, 8),
                                                                                ("poisson", 4, 8), ("poisson3d", 2, 5))]


@pytest.mark.parametrize("group", range(10))
def test_one_hundred_more_random_programs(native, oracle, group):
    """100 programs of tools/fuzz_more.py's stream (2D real, 2D complex with Robin columns, 3D; default thresholds — the kernels a
    user gets), alternating `with jacobi` semantics: every field of every level bit for bit after run_ops."""
    for n, (kind, lo, hi, seed) in enumerate(MORE[10 * group:10 * group + 10]):
        semantics = ("textbook", "exastencils-v1.1")[(seed + n) % 2]
        if kind == "poisson3d":
            pd = problems.poisson_3d(lo, hi)
        elif kind == "poisson":
            pd = problems.poisson_2d(lo, hi)
        else:
            pd = problems.helmholtz_2d(lo, hi, k=0.625 * (1 << hi))
        pd = with_semantics(pd, semantics)
        rng = np.random.default_rng(31 * seed + lo)
        ops = random_program(rng, 3 if kind == "poisson3d" else 2, lo, hi, 28)
        hg, ho = native.create(pd), oracle.create(pd)
        for L in range(lo, hi + 1):
            shape = hg.field_shape(L)
            for f in (abi.F_SOL, abi.F_RHS, abi.F_RES):
                a = rng.standard_normal(shape) + (1j * rng.standard_normal(shape) if kind == "helmholtz" else 0.0)
                hg.set_field(L, f, a)
                ho.set_field(L, f, a)
        hg.run_ops(ops)
        ho.run_ops(ops)
        for L in range(lo, hi + 1):
            for f in (abi.F_SOL, abi.F_RHS, abi.F_RES):
                got, want = hg.get_field(L, f), ho.get_field(L, f)
                assert np.array_equal(got, want, equal_nan=True), (kind, lo, hi, semantics, seed, L, f)
        hg.close()
        ho.close()
