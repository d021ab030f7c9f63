"""GPU parity tests proper: the HIP backend (through the C-ABI, csrc/libevomg.so) against the CPU oracle
on the same seeded inputs.

Bars
 * pointwise kernels (smoothers, residual, restriction, prolongation, zero, copy, Robin BC): BIT-EXACT
   (identical operation order, both sides built with -ffp-contract=off);
 * reductions / Krylov coarse solves / whole solves: floating-point order of the sums differs between the
   wave-shuffle tree and the oracle's row-wise sum -> relative tolerance 1e-10 on residual norms (the
   tolerance north_star states), identical iteration counts.
"""
import dataclasses
import numpy as np
import pytest

from evostencils_amd import abi, problems
import cycles
from helpers import ops_for_case

pytestmark = pytest.mark.gpu

REL_TOL = 1e-10


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()  # raises if the HIP library is missing: no fallback
    return nat


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as orc
    return orc


def make_pair(native, oracle, pd, parameters=None):
    return native.create(pd, 0, parameters), oracle.create(pd, parameters)


def random_field(rng, shape, is_complex):
    a = rng.standard_normal(shape)
    if is_complex:
        a = a + 1j * rng.standard_normal(shape)
    return a


def zero_boundary(a):
    a = a.copy()
    for d in range(a.ndim):
        sl = [slice(None)] * a.ndim
        sl[d] = 0
        a[tuple(sl)] = 0
        sl[d] = -1
        a[tuple(sl)] = 0
    return a


PROBLEMS = {
    "poisson2d": lambda: problems.poisson_2d(3, 6),
    "poisson3d": lambda: problems.poisson_3d(2, 4),
    "helmholtz2d": lambda: problems.helmholtz_2d(3, 6, k=20.0),
}


def load_random_state(hg, ho, pd, rng):
    """Same random contents in every field of every level on both sides."""
    for L in range(pd.min_level, pd.max_level + 1):
        shape = hg.field_shape(L)
        fields = [abi.F_SOL, abi.F_RHS, abi.F_RES]
        if L == pd.max_level:
            fields.append(abi.F_CORR)
        if L == pd.min_level:
            fields.append(abi.F_CGS_RHS)
        for f in fields:
            a = random_field(rng, shape, hg.is_complex)
            if pd.bc_kind == abi.BC_HELMHOLTZ_ROBIN:
                a = zero_boundary(a)
            hg.set_field(L, f, a)
            ho.set_field(L, f, a)


def pointwise_ops(pd):
    d = pd.dimension
    lo, hi = pd.min_level, pd.max_level
    mid = hi - 1
    one = (1, 1, 1)
    ops = [
        ("jacobi", abi.make_op(abi.OP_SMOOTH, hi, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 0.8), (hi, abi.F_SOL)),
        ("jacobi_coarse", abi.make_op(abi.OP_SMOOTH, mid, abi.F_CORR, 0, abi.F_RHS, abi.PART_SINGLE, one, 1.3), (mid, abi.F_SOL)),
        ("rbgs", abi.make_op(abi.OP_SMOOTH, hi, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, 1.15), (hi, abi.F_SOL)),
        ("rbgs_coarse", abi.make_op(abi.OP_SMOOTH, lo, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, 0.6), (lo, abi.F_SOL)),
        ("residual", abi.make_op(abi.OP_RESIDUAL, hi, abi.F_RES, abi.F_SOL, abi.F_RHS), (hi, abi.F_RES)),
        ("richardson", abi.make_op(abi.OP_RICHARDSON, hi, abi.F_SOL, abi.F_SOL, abi.F_RHS, omega=0.3 / abs(complex(pd.mg_stencil(hi, pd.parameters)[0]))), (hi, abi.F_SOL)),
        ("restrict", abi.make_op(abi.OP_RESTRICT, mid, abi.F_RHS, abi.F_RES), (mid, abi.F_RHS)),
        ("prolong_add", abi.make_op(abi.OP_PROLONG_ADD, hi, abi.F_SOL, abi.F_CORR, omega=0.95), (hi, abi.F_SOL)),
        ("prolong_set", abi.make_op(abi.OP_PROLONG_SET, hi, abi.F_CORR, abi.F_CORR), (hi, abi.F_CORR)),
        ("zero", abi.make_op(abi.OP_SET_ZERO, mid, abi.F_SOL), (mid, abi.F_SOL)),
        ("copy", abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS), (lo, abi.F_CGS_RHS)),
    ]
    shapes2 = [(2, 1), (1, 2), (2, 2), (3, 2), (1, 6), (4, 2), (8, 1), (1, 3)]
    shapes3 = [(2, 1, 1), (1, 2, 1), (1, 1, 2), (2, 2, 1), (2, 2, 2), (1, 2, 3), (4, 1, 2)]
    for b in (shapes2 if d == 2 else shapes3):
        blk = tuple(b) + (1,) * (3 - len(b))
        ops.append((f"block{b}", abi.make_op(abi.OP_SMOOTH, hi, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, blk, 0.7), (hi, abi.F_SOL)))
    return ops


@pytest.mark.parametrize("pname", sorted(PROBLEMS))
def test_pointwise_kernels_are_bit_exact(native, oracle, pname):
    pd = PROBLEMS[pname]()
    hg, ho = make_pair(native, oracle, pd)
    rng = np.random.default_rng(1234)
    for name, op, (lvl, fld) in pointwise_ops(pd):
        load_random_state(hg, ho, pd, rng)
        hg.run_ops([op])
        ho.run_ops([op])
        got, want = hg.get_field(lvl, fld), ho.get_field(lvl, fld)
        assert np.array_equal(got, want), f"{pname}:{name}: max diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("pname", sorted(PROBLEMS))
def test_smoother_sequences_stay_bit_exact(native, oracle, pname):
    """Ping-pong buffers, colour order and boundary handling over several consecutive sweeps."""
    pd = PROBLEMS[pname]()
    hg, ho = make_pair(native, oracle, pd)
    rng = np.random.default_rng(99)
    load_random_state(hg, ho, pd, rng)
    hi = pd.max_level
    one = (1, 1, 1)
    blk = (2, 1, 1) if pd.dimension == 2 else (2, 2, 1)
    seq = [abi.make_op(abi.OP_SMOOTH, hi, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 0.8),
           abi.make_op(abi.OP_SMOOTH, hi, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, 1.1),
           abi.make_op(abi.OP_SMOOTH, hi, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, blk, 0.9),
           abi.make_op(abi.OP_SMOOTH, hi, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 0.5),
           abi.make_op(abi.OP_SMOOTH, hi, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, 1.0),
           abi.make_op(abi.OP_RESIDUAL, hi, abi.F_RES, abi.F_SOL, abi.F_RHS)]
    hg.run_ops(seq)
    ho.run_ops(seq)
    assert np.array_equal(hg.get_field(hi, abi.F_SOL), ho.get_field(hi, abi.F_SOL))
    assert np.array_equal(hg.get_field(hi, abi.F_RES), ho.get_field(hi, abi.F_RES))


@pytest.mark.parametrize("pname", sorted(PROBLEMS))
def test_norm_and_coarse_solver_within_tolerance(native, oracle, pname):
    pd = PROBLEMS[pname]()
    hg, ho = make_pair(native, oracle, pd)
    rng = np.random.default_rng(7)
    load_random_state(hg, ho, pd, rng)
    for L in range(pd.min_level, pd.max_level + 1):
        a, b = hg.norm(L, abi.F_RHS), ho.norm(L, abi.F_RHS)
        assert abs(a - b) <= 1e-13 * b
    lo = pd.min_level
    zero = abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR)
    solve = abi.make_op(abi.OP_COARSE_SOLVE, lo)
    res = abi.make_op(abi.OP_RESIDUAL, lo, abi.F_RES, abi.F_CORR, abi.F_CGS_RHS)
    hg.run_ops([zero, solve, res])
    ho.run_ops([zero, solve, res])
    # canonical reduction order on both sides (oracle FN(dot_canon)): the Krylov solve is bit-identical
    assert np.array_equal(hg.get_field(lo, abi.F_CORR), ho.get_field(lo, abi.F_CORR))
    # both must actually have solved the system to the configured reduction
    tol = pd.cgs_target_res_reduction
    assert hg.norm(lo, abi.F_RES) <= 50 * tol * ho.norm(lo, abi.F_CGS_RHS)


def compare_solves(og, oo, same_iterations=True):
    assert og.status == oo.status
    if same_iterations:
        assert og.n_iterations == oo.n_iterations
        assert len(og.history) == len(oo.history)
        rel = np.abs(og.history - oo.history) / oo.history
        assert rel.max() <= REL_TOL, f"residual history differs: max rel {rel.max():.3e}"
    assert abs(og.final_res_norm - oo.final_res_norm) <= REL_TOL * oo.final_res_norm + 1e-300 or not same_iterations


@pytest.mark.parametrize("name", sorted(cycles.CASES))
def test_whole_solves_match_the_oracle(native, oracle, name):
    """Every golden-string candidate: lowering -> op program -> solve on both sides."""
    problem, lo, hi, _ = cycles.CASES[name]
    gen, ops = ops_for_case(name)
    pd = problems.BY_NAME[problem](lo, hi)
    if name == "elasticity2d_L5_mixed_shapes":
        # one block shape per unknown: accepted only where the sweep's result is never used (include/evomg.h: block_other)
        import dataclasses
        pd = dataclasses.replace(pd, jacobi_semantics=abi.JACOBI_SEMANTICS["exastencils-v1.1"])
    hg, ho = make_pair(native, oracle, pd)
    og = hg.evaluate(ops, 1, 0, 128)
    oo = ho.evaluate(ops, 1, 0, 128)
    compare_solves(og, oo)
    # every field-updating kernel incl. the Krylov coarse solver follows the oracle's operation order
    # => the converged solution is bit-identical; only the (non-feedback) outer norm differs in sum order
    assert np.array_equal(hg.get_field(hi, abi.F_SOL), ho.get_field(hi, abi.F_SOL))


def test_helmholtz_pbicgstab_matches_the_oracle(native, oracle):
    """Reference V(2,1) RB-GS w=0.6 cycle of Helmholtz exa3:202-272 as preconditioner, k*h = 0.625."""
    from test_helmholtz_oracle import reference_vcycle_ops
    lo, hi = 3, 6
    pd = problems.helmholtz_2d(lo, hi, k=0.625 * (1 << hi))
    hg, ho = make_pair(native, oracle, pd)
    ops = reference_vcycle_ops(lo, hi)
    og = hg.evaluate(ops, 1, 0, 256)
    oo = ho.evaluate(ops, 1, 0, 256)
    assert og.status == oo.status == abi.SOLVE_CONVERGED
    assert og.n_iterations == oo.n_iterations
    rel = np.abs(og.history - oo.history) / oo.history
    # north_star's tolerance; in fact every dot of the outer recurrence is summed in one canonical order on both sides
    # (oracle FN(dot) restates evomg_dot_partial / evomg_dot_final), so the two trajectories are the same bit for bit
    # (tools/debug_helmholtz.py: 0.0 at levels 3..6 and 5..9)
    assert rel.max() <= REL_TOL
    sg, so = hg.get_field(hi, abi.F_CORR), ho.get_field(hi, abi.F_CORR)
    assert np.abs(sg - so).max() <= REL_TOL * np.abs(so).max()


@pytest.mark.parametrize("lo,hi", [(3, 6), (5, 9)])
def test_outer_bicgstab_fused_passes_do_not_change_a_bit(native, oracle, lo, hi, monkeypatch):
    """The preconditioned BiCGStab loop with its fused passes (right-hand side of the preconditioner read in place, both
    updates of p in one pass, A u with its dot products, residual update with norm and next rho) against the same solve
    with EVOMG_NO_OUTER_FUSE, and with only round 5's folds switched off (EVOMG_NO_OUTER_FOLDS: beta formed by a launch of its
    own, the Robin columns of the iterate by a boundary pass behind the update): identical histories and solutions; and the
    oracle's history to 1e-10."""
    from test_helmholtz_oracle import reference_vcycle_ops
    pd = problems.helmholtz_2d(lo, hi, k=0.625 * (1 << hi))
    ops = reference_vcycle_ops(lo, hi)
    results = []
    for knob in (None, "EVOMG_NO_OUTER_FUSE", "EVOMG_NO_OUTER_FOLDS"):
        if knob:
            monkeypatch.setenv(knob, "1")
        h = native.create(pd)
        out = h.evaluate(ops, 1, 0, 1024)
        results.append((out, h.get_field(hi, abi.F_CORR), h.get_field(hi, abi.F_RES)))
        h.close()
        if knob:
            monkeypatch.delenv(knob)
    (a, xa, ra) = results[0]
    for (b, xb, rb) in results[1:]:
        assert a.status == b.status and a.n_iterations == b.n_iterations and a.n_iterations > 3
        assert np.array_equal(a.history, b.history) and np.array_equal(xa, xb) and np.array_equal(ra, rb)
    if hi <= 6:
        oo = oracle.create(pd).evaluate(ops, 1, 0, 1024)
        assert oo.n_iterations == a.n_iterations
        assert (np.abs(a.history - oo.history) / oo.history).max() <= REL_TOL


def test_evolved_helmholtz_cycle(native, oracle):
    """The frozen evolved Helmholtz preconditioner (tests/test_helmholtz_oracle.py: two V(2,1) corrections per
    application): against the oracle at 65^2, and at 257^2 (k = 160) by the invariants of the returned solution —
    converged to 1e-7 in fewer outer iterations than the hand-written cycle needs, true residual of A u = f, Robin
    columns."""
    from test_helmholtz_oracle import evolved_double_cycle_ops, reference_vcycle_ops
    lo, hi = 3, 6
    gen, ops = evolved_double_cycle_ops(lo, hi, 0.625 * (1 << hi))
    hg, ho = make_pair(native, oracle, gen.problem)
    og, oo = hg.evaluate(ops, 1, 0, 256), ho.evaluate(ops, 1, 0, 256)
    assert og.status == oo.status == abi.SOLVE_CONVERGED and og.n_iterations == oo.n_iterations
    assert (np.abs(og.history - oo.history) / oo.history).max() <= REL_TOL
    lo, hi = 4, 8
    k = 0.625 * (1 << hi)
    gen, ops = evolved_double_cycle_ops(lo, hi, k)
    h = native.create(gen.problem)
    out = h.evaluate(ops, 1, 0, 0)
    ref = h.evaluate(reference_vcycle_ops(lo, hi), 1, 0, 0)
    assert out.status == ref.status == abi.SOLVE_CONVERGED and out.n_iterations < ref.n_iterations      # 1211 vs 1594
    out = h.evaluate(ops, 1, 0, 0)
    u = h.get_field(hi, abi.F_CORR)
    n = (1 << hi) + 1
    hh = 1.0 / (n - 1)
    f = gen.problem.right_hand_side(hi)
    Au = (4.0 / hh ** 2 - k ** 2) * u[1:-1, 1:-1] - (u[1:-1, :-2] + u[1:-1, 2:] + u[:-2, 1:-1] + u[2:, 1:-1]) / hh ** 2
    r = f[1:-1, 1:-1] - Au
    assert np.sqrt(abs(np.sum(r * r))) / out.init_res_norm < 1e-6
    assert np.allclose(u[1:-1, 0], u[1:-1, 1] / (1 - 1j * k * hh), rtol=1e-12, atol=0)
    assert np.allclose(u[1:-1, -1], u[1:-1, -2] / (1 - 1j * k * hh), rtol=1e-12, atol=0)
    h.close()


def test_evaluation_samples_and_reset(native):
    """Fields are restored before every sample: repeated evaluations give identical numbers."""
    pd = problems.poisson_2d(3, 7)
    h = native.create(pd)
    gen, ops = ops_for_case("poisson2d_L7_v21_rbgs")
    pd = problems.poisson_2d(4, 7)
    h = native.create(pd)
    a = h.evaluate(ops, 3, 0, 128)
    b = h.evaluate(ops, 1, 0, 128)
    assert a.n_iterations == b.n_iterations and np.array_equal(a.history, b.history)
    assert a.time_ms > 0


def test_large_grid_properties(native):
    """Sizes the oracle would not finish quickly: size-independent properties on the GPU alone —
    textbook V(1,1) RB-GS convergence, O(h^2) error against the analytic solution, and residual
    reduction to 1e-12 (2D level 10: 1025^2; 3D level 6: 65^3)."""
    from test_oracle_known_answers import simple_vcycle, exact_2d
    errs = {}
    for hi in (9, 10):
        pd = problems.poisson_2d(4, hi)
        h = native.create(pd)
        out = h.evaluate(simple_vcycle(4, hi), 1, 0, 128)
        assert out.status == abi.SOLVE_CONVERGED and out.conv_factor < 0.1
        errs[hi] = np.abs(h.get_field(hi, abi.F_SOL) - exact_2d(hi)).max()
        h.close()
    assert 3.7 < errs[9] / errs[10] < 4.3
    pd = problems.poisson_3d(2, 6)
    h = native.create(pd)
    out = h.evaluate(simple_vcycle(2, 6, pre=2, post=2, om=0.8, part=abi.PART_SINGLE, block=(2, 1, 1)), 1, 0, 128)
    assert out.status == abi.SOLVE_CONVERGED
    n = 65
    x = np.arange(n) / (n - 1)
    exact = x[None, None, :] ** 2 - 0.5 * x[None, :, None] ** 2 - 0.5 * x[:, None, None] ** 2
    assert np.abs(h.get_field(6, abi.F_SOL) - exact).max() < 1e-10


def test_rejects_malformed_programs(native):
    h = native.create(problems.poisson_2d(3, 5))
    rc, _ = h.try_evaluate([abi.make_op(abi.OP_SMOOTH, 9, abi.F_SOL, 0, abi.F_RHS)], 1, 0, 0)
    assert rc == abi.ERR_INVALID
    rc, _ = h.try_evaluate([abi.make_op(abi.OP_SMOOTH, 5, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (2, 1, 1))], 1, 0, 0)
    assert rc == abi.ERR_UNSUPPORTED
    rc, _ = h.try_evaluate([abi.make_op(abi.OP_COARSE_SOLVE, 4)], 1, 0, 0)
    assert rc == abi.ERR_INVALID
    rc, _ = h.try_evaluate([abi.make_op(abi.OP_COPY, 5, abi.F_CGS_RHS, abi.F_RHS)], 1, 0, 0)
    assert rc == abi.ERR_INVALID   # gen_rhs_u exists on the coarsest level only


@pytest.mark.parametrize("lo", [1, 2, 3, 4, 5])
def test_one_wave_cg_and_its_folded_copy_and_zero(native, oracle, lo, monkeypatch):
    """The classical form of the coarse CG (cgs_kind CGS_CG_CLASSICAL; the default kind's kernels: tests/test_cg_variants_gpu.py).
    2D grids up to 33^2: the whole CG in one wave (evomg_cg_wave2d), and `gen_rhs = RHS; gen_error = 0; gen_mgCycle@min()`
    as ONE launch (the kernel copies the right-hand side and starts from x = 0 without reading x) — against the three
    separate ops, the global-memory kernel and the oracle: solution, residual and the copied right-hand side bit for bit.
    With a non-zero BOUNDARY on x (a test-only state) the fold must not happen: r = b - A x then sees those values."""
    pd = dataclasses.replace(problems.poisson_2d(lo, lo + 1), cgs_kind=abi.CGS_CG_CLASSICAL)
    n = (1 << lo) + 1
    rng = np.random.default_rng(50 + lo)
    rhs = rng.standard_normal((n, n))
    copy = abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS)
    zero = abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR)
    solve = abi.make_op(abi.OP_COARSE_SOLVE, lo)
    # x0 = None: the error field as a cycle finds it (never touched from outside: zero, boundary known to be zero) — the
    # fold happens; x0 random incl. its boundary: set_field marks the boundary as unknown, no fold
    for x0 in (None, rng.standard_normal((n, n))):
        ho = oracle.create(pd)
        ho.set_field(lo, abi.F_RHS, rhs)
        if x0 is not None:
            ho.set_field(lo, abi.F_SOL, x0)
        ho.run_ops([copy, zero, solve])
        want = [ho.get_field(lo, f) for f in (abi.F_SOL, abi.F_RES, abi.F_CGS_RHS)]
        for programs, knob in (([[copy, zero, solve]], None), ([[copy], [zero], [solve]], None), ([[copy, zero, solve]], "EVOMG_NO_FUSE"),
                               ([[copy, zero, solve]], "EVOMG_NO_REG_KRYLOV")):
            if knob:
                monkeypatch.setenv(knob, "1")
            h = native.create(pd)
            h.set_field(lo, abi.F_RHS, rhs)
            if x0 is not None:
                h.set_field(lo, abi.F_SOL, x0)
            for prog in programs:
                h.run_ops(prog)
            got = [h.get_field(lo, f) for f in (abi.F_SOL, abi.F_RES, abi.F_CGS_RHS)]
            h.close()
            if knob:
                monkeypatch.delenv(knob)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]), (programs, knob)
            assert np.array_equal(got[1][1:-1, 1:-1], want[1][1:-1, 1:-1])


@pytest.mark.parametrize("dim,lo", [(2, 8), (3, 6)])
def test_large_coarsest_grids_run_cg_as_launches(native, oracle, dim, lo):
    """257^2 / 65^3 and up: the coarse CG as three many-unit launches per iteration with the scalars in a device-side state
    (kcg_*), dot products in the blocked order with fused accumulation that the oracle restates: solution and residual bit
    for bit, and two orders of magnitude faster than one workgroup (tools/coarse_big.py)."""
    pd = problems.poisson_2d(lo, lo + 1) if dim == 2 else problems.poisson_3d(lo, lo + 1)
    rng = np.random.default_rng(lo * 10 + dim)
    shape = ((1 << lo) + 1,) * dim
    rhs = rng.standard_normal(shape)
    x0 = zero_boundary(rng.standard_normal(shape)) * 1e-3
    ops = [abi.make_op(abi.OP_COARSE_SOLVE, lo)]
    h = native.create(pd)
    ho = oracle.create(pd)
    for hh in (h, ho):
        hh.set_field(lo, abi.F_CGS_RHS, rhs)
        hh.set_field(lo, abi.F_CORR, x0)
        hh.run_ops(ops)
    inner = (slice(1, -1),) * dim
    assert np.array_equal(h.get_field(lo, abi.F_CORR), ho.get_field(lo, abi.F_CORR))
    assert np.array_equal(h.get_field(lo, abi.F_RES)[inner], ho.get_field(lo, abi.F_RES)[inner])
    assert not np.array_equal(h.get_field(lo, abi.F_CORR), x0)
    h.close()
    ho.close()


@pytest.mark.parametrize("dim,lo", [(2, 3), (2, 5), (2, 6), (2, 7), (3, 2), (3, 3), (3, 4), (3, 5)])
def test_coarse_cg_variants_are_bit_identical(native, oracle, dim, lo, monkeypatch):
    """The register-resident CG (2D) / the eight-workgroup CG (33^3), the global-memory single-workgroup CG
    and the oracle follow one summation order: all three give the same bits (129^2 = the C2 coarsest grid up to round 4).
    The classical form: cgs_kind CGS_CG_CLASSICAL (the default kind's single-reduction kernels: tests/test_cg_variants_gpu.py)."""
    pd = problems.poisson_2d(lo, lo + 1) if dim == 2 else problems.poisson_3d(lo, lo + 1)
    pd = dataclasses.replace(pd, cgs_kind=abi.CGS_CG_CLASSICAL)
    rng = np.random.default_rng(lo * 10 + dim)
    shape = ((1 << lo) + 1,) * dim
    rhs = rng.standard_normal(shape)
    x0 = zero_boundary(rng.standard_normal(shape)) * 1e-3
    ops = [abi.make_op(abi.OP_COARSE_SOLVE, lo)]
    results = []
    for no_lds in (False, True):
        if no_lds:
            monkeypatch.setenv("EVOMG_NO_REG_KRYLOV", "1")
        else:
            monkeypatch.delenv("EVOMG_NO_REG_KRYLOV", raising=False)
        h = native.create(pd)
        h.set_field(lo, abi.F_CGS_RHS, rhs)
        h.set_field(lo, abi.F_CORR, x0)
        h.run_ops(ops)
        results.append((h.get_field(lo, abi.F_CORR), h.get_field(lo, abi.F_RES)))
        h.close()
    ho = oracle.create(pd)
    ho.set_field(lo, abi.F_CGS_RHS, rhs)
    ho.set_field(lo, abi.F_CORR, x0)
    ho.run_ops(ops)
    want_x, want_r = ho.get_field(lo, abi.F_CORR), ho.get_field(lo, abi.F_RES)
    for got_x, got_r in results:
        assert np.array_equal(got_x, want_x)
        assert np.array_equal(got_r[(slice(1, -1),) * dim], want_r[(slice(1, -1),) * dim])


@pytest.mark.parametrize("pname,level", [("poisson2d", 8), ("poisson2d", 9), ("helmholtz2d", 8)])
@pytest.mark.parametrize("rows,pf", [(32, 0), (8, 1), (64, 0), (13, 1)])
def test_marching_strip_kernels_match_generic_kernels(native, pname, level, rows, pf, monkeypatch):
    """The DPP marching-strip kernels (Jacobi, fused two-colour RB-GS, residual) against the plain
    thread-per-node kernels (themselves bit-exact vs the oracle) on grids with several strips and chunks,
    for several chunk heights, with and without non-temporal right-hand-side loads."""
    pd = problems.poisson_2d(level - 1, level) if pname == "poisson2d" else problems.helmholtz_2d(level - 1, level, k=30.0)
    rng = np.random.default_rng(level * 100 + rows)
    shape = ((1 << level) + 1,) * 2
    cx = pname != "poisson2d"
    x0 = random_field(rng, shape, cx)
    f0 = random_field(rng, shape, cx)
    r0 = random_field(rng, shape, cx)
    if cx:
        x0, f0, r0 = zero_boundary(x0), zero_boundary(f0), zero_boundary(r0)
    one = (1, 1, 1)
    seq = [abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, 1.15),
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 0.8),
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, 0.6),
           abi.make_op(abi.OP_RESIDUAL, level, abi.F_RES, abi.F_SOL, abi.F_RHS),
           # residual immediately followed by its restriction: runs as the fused strip kernel
           abi.make_op(abi.OP_RESTRICT, level - 1, abi.F_RHS, abi.F_RES),
           abi.make_op(abi.OP_SMOOTH, level - 1, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, 1.0),
           abi.make_op(abi.OP_PROLONG_ADD, level, abi.F_SOL, abi.F_SOL, omega=0.85),
           abi.make_op(abi.OP_PROLONG_SET, level, abi.F_CORR, abi.F_SOL),
           # prolongation + correction immediately followed by a point smoother: one fused sweep (real problems)
           abi.make_op(abi.OP_PROLONG_ADD, level, abi.F_SOL, abi.F_SOL, omega=1.1),
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, 0.9),
           abi.make_op(abi.OP_PROLONG_ADD, level, abi.F_SOL, abi.F_SOL, omega=0.7),
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 0.8)]
    out = []
    for strip in (True, False):
        if strip:
            monkeypatch.delenv("EVOMG_NO_STRIP", raising=False)
            monkeypatch.setenv("EVOMG_STRIP_ROWS", str(rows))
            monkeypatch.setenv("EVOMG_STRIP_NT", str(pf))   # pf: non-temporal rhs loads on/off
        else:
            monkeypatch.setenv("EVOMG_NO_STRIP", "1")
        h = native.create(pd)
        h.set_field(level, abi.F_SOL, x0)
        h.set_field(level, abi.F_RHS, f0)
        h.set_field(level, abi.F_RES, r0)
        h.run_ops(seq)
        out.append((h.get_field(level, abi.F_SOL), h.get_field(level, abi.F_RES), h.get_field(level - 1, abi.F_RHS),
                    h.get_field(level - 1, abi.F_SOL), h.get_field(level, abi.F_CORR)))
        h.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("level,planes,waves", [(4, 4, 0), (5, 8, 4), (6, 64, 8), (6, 12, 4), (7, 16, 8), (7, 0, 0), (7, 8, 4)])
def test_marching_tile_kernels_3d_match_generic_kernels(native, level, planes, waves, monkeypatch):
    """3D marching-tile kernels (lane = column pair, TY rows per wave, four-plane register ring along z) against the
    thread-per-node kernels (bit-exact vs the oracle) on grids with partial strips, partial y tiles and several z
    chunks."""
    pd = problems.poisson_3d(level - 1, level)
    rng = np.random.default_rng(level * 100 + planes)
    shape = ((1 << level) + 1,) * 3
    x0, f0, r0 = (random_field(rng, shape, False) for _ in range(3))
    one = (1, 1, 1)
    seq = [abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 0.8),
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, 1.1),
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 0.6),
           abi.make_op(abi.OP_RESIDUAL, level, abi.F_RES, abi.F_SOL, abi.F_RHS),
           abi.make_op(abi.OP_RESTRICT, level - 1, abi.F_RHS, abi.F_RES),
           abi.make_op(abi.OP_SMOOTH, level - 1, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 1.0),
           abi.make_op(abi.OP_SMOOTH, level - 1, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, 1.0),
           abi.make_op(abi.OP_PROLONG_ADD, level, abi.F_SOL, abi.F_SOL, omega=0.85),
           abi.make_op(abi.OP_PROLONG_SET, level, abi.F_CORR, abi.F_SOL),
           # two-cell block-Jacobi shapes run as modes of the LDS-halo tile kernel (evomg_tile3d.h)
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 1, 1), 0.7),
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1, 2, 1), 0.9),
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 1, 1), 1.0),
           # the same Jacobi-type smoother twice in a row: ONE pass of evomg_tile3d_twice (temporal fusion)
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 1, 1), 0.8),
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 1, 1), 0.8),
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 0.7),
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 0.7),
           abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 0.7),
           abi.make_op(abi.OP_RESIDUAL, level, abi.F_RES, abi.F_SOL, abi.F_RHS)]
    out = []
    for strip in (True, False):
        if strip:
            monkeypatch.delenv("EVOMG_NO_STRIP", raising=False)
            monkeypatch.setenv("EVOMG_STRIP3D_MIN_N", "9")
            monkeypatch.setenv("EVOMG_STRIP_PLANES", str(planes))
            monkeypatch.setenv("EVOMG_TILE3D_WAVES", str(waves))
        else:
            monkeypatch.setenv("EVOMG_NO_STRIP", "1")
        h = native.create(pd)
        h.set_field(level, abi.F_SOL, x0)
        h.set_field(level, abi.F_RHS, f0)
        h.set_field(level, abi.F_RES, r0)
        h.run_ops(seq)
        out.append((h.get_field(level, abi.F_SOL), h.get_field(level, abi.F_RES), h.get_field(level - 1, abi.F_RHS),
                    h.get_field(level - 1, abi.F_SOL), h.get_field(level, abi.F_CORR)))
        h.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("smoother,pre,post,planes,waves", [("block211", 2, 2, 0, 0), ("jacobi", 2, 2, 8, 4), ("rbgs", 1, 1, 0, 8),
                                                              ("block211", 1, 2, 12, 8)])
def test_sweep_fusions_3d_do_not_change_a_bit(native, smoother, pre, post, planes, waves, monkeypatch):
    """3D whole solves (129^3, three levels; the tiles forced onto 65^3 too) with the interpreter's fusions — two sweeps in
    one pass (evomg_tile3d_twice), residual + restriction without the residual field (evomg_tile3d_residual_restrict),
    residual + norm (TILE_RESNORM) — against the same solves with EVOMG_NO_FUSE: identical residual histories,
    solutions, residuals and coarse right-hand sides."""
    from test_full_size_gpu import vcycle
    lo, hi = 5, 7
    pd = problems.poisson_3d(lo, hi)
    part, block, om = {"block211": (abi.PART_SINGLE, (2, 1, 1), 0.8), "jacobi": (abi.PART_SINGLE, (1, 1, 1), 0.8),
                       "rbgs": (abi.PART_RED_BLACK, (1, 1, 1), 1.1)}[smoother]
    ops = vcycle(lo, hi, pre, post, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, part, block, om))
    monkeypatch.setenv("EVOMG_STRIP3D_MIN_N", "33")
    monkeypatch.setenv("EVOMG_STRIP_PLANES", str(planes))
    monkeypatch.setenv("EVOMG_TILE3D_WAVES", str(waves))
    results = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("EVOMG_NO_FUSE", "1")
        else:
            monkeypatch.delenv("EVOMG_NO_FUSE", raising=False)
        h = native.create(pd)
        out = h.evaluate(ops, 1, 0, 128)
        results.append((out, h.get_field(hi, abi.F_SOL), h.get_field(hi, abi.F_RES), h.get_field(hi - 1, abi.F_RHS),
                        h.get_field(lo, abi.F_RHS)))
        h.close()
    (a, *fa), (b, *fb) = results
    assert a.status == b.status and a.n_iterations == b.n_iterations and a.n_iterations > 3
    assert np.array_equal(a.history, b.history)
    for x, y in zip(fa, fb):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("smoother,max_n", [("rbgs", 129), ("jacobi", 129), ("rbgs", 33)])
def test_coarse_level_batches_do_not_change_a_bit(native, smoother, max_n, monkeypatch):
    """EVOMG_BATCH=1: consecutive ops on levels up to max_n^2 run as micro-ops of ONE single-workgroup launch
    (evomg_coarse_batch). Whole solves against the default interpreter: identical histories and fields."""
    from test_full_size_gpu import vcycle
    lo, hi = 4, 8
    pd = problems.poisson_2d(lo, hi)
    part, om = (abi.PART_RED_BLACK, 1.1) if smoother == "rbgs" else (abi.PART_SINGLE, 0.8)
    ops = vcycle(lo, hi, 2, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, part, (1, 1, 1), om))
    results = []
    for batched in (True, False):
        if batched:
            monkeypatch.setenv("EVOMG_BATCH", "1")
            monkeypatch.setenv("EVOMG_BATCH_MAX_N", str(max_n))
        else:
            monkeypatch.delenv("EVOMG_BATCH", raising=False)
        h = native.create(pd)
        out = h.evaluate(ops, 1, 0, 128)
        results.append((out, h.get_field(hi, abi.F_SOL), h.get_field(hi - 2, abi.F_SOL), h.get_field(hi - 2, abi.F_RHS),
                        h.get_field(lo, abi.F_RHS)))
        h.close()
    (a, *fa), (b, *fb) = results
    assert a.status == b.status and a.n_iterations == b.n_iterations and a.n_iterations > 3
    assert np.array_equal(a.history, b.history)
    for x, y in zip(fa, fb):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("smoother,pre,post", [("rbgs", 1, 1), ("rbgs", 2, 2), ("jacobi", 1, 2), ("rbgs", 0, 1), ("jacobi", 2, 0)])
def test_sweep_fusions_do_not_change_a_bit(native, smoother, pre, post, monkeypatch):
    """Whole solves with every optimisation of the interpreter (residual+restriction, prolongation+smoother,
    smoother+residual+restriction, norm on the last sweep, speculative first sweep, power-of-two reciprocal) against
    the same solves with all of them switched off: identical residual histories, iteration counts, solution and
    residual fields, and coarse right-hand sides."""
    from test_full_size_gpu import vcycle
    lo, hi = 5, 9
    pd = problems.poisson_2d(lo, hi)
    part, om = (abi.PART_RED_BLACK, 1.1) if smoother == "rbgs" else (abi.PART_SINGLE, 0.8)
    ops = vcycle(lo, hi, pre, post, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, part, (1, 1, 1), om))
    knobs = ["EVOMG_NO_FUSE", "EVOMG_NO_TAIL_NORM", "EVOMG_NO_TAIL_RESTRICT", "EVOMG_NO_SPECULATION",
             "EVOMG_NO_POW2_RECIPROCAL", "EVOMG_NO_ZERO_ELISION"]
    results = []
    for plain in (False, True):
        for k in knobs:
            if plain:
                monkeypatch.setenv(k, "1")
            else:
                monkeypatch.delenv(k, raising=False)
        h = native.create(pd)
        out = h.evaluate(ops, 1, 0, 128)
        results.append((out, h.get_field(hi, abi.F_SOL), h.get_field(hi, abi.F_RES), h.get_field(hi - 1, abi.F_RHS),
                        h.get_field(lo, abi.F_RHS)))
        h.close()
    (a, *fa), (b, *fb) = results
    assert a.status == b.status and a.n_iterations == b.n_iterations
    assert np.array_equal(a.history, b.history)
    for x, y in zip(fa, fb):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("post", ["jacobi", "block211"])
def test_lost_jacobi_sweeps_3d_keep_the_coarse_grid_correction(native, oracle, post, monkeypatch):
    """EVOMG_JACOBI_EXASTENCILS_V11 on the 3D tiles (forced on from 33^3): PROLONG_ADD followed by two Single-partitioned
    sweeps must not become one double sweep — its only output is the array the semantics discard, and the correction
    folded into its loads would be discarded with it. RB-GS pre-smoothing keeps the cycle convergent; the two lost
    post-sweeps are launched and change nothing. GPU vs oracle (same semantics): identical iteration counts, histories
    within 1e-10, solution bit-identical; and identical to the same solve with EVOMG_NO_FUSE."""
    import dataclasses
    lo, hi = 4, 6
    pd = dataclasses.replace(problems.poisson_3d(lo, hi), jacobi_semantics=abi.JACOBI_EXASTENCILS_V11)
    block = (2, 1, 1) if post == "block211" else (1, 1, 1)
    rb = lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.0)
    lost = lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, block, 0.8)
    ops = []
    for L in range(hi, lo, -1):
        ops += [rb(L), rb(L), abi.make_op(abi.OP_RESIDUAL, L, abi.F_RES, abi.F_SOL, abi.F_RHS),
                abi.make_op(abi.OP_RESTRICT, L - 1, abi.F_RHS, abi.F_RES)]
        if L - 1 > lo:
            ops.append(abi.make_op(abi.OP_SET_ZERO, L - 1, abi.F_SOL))
    ops += [abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS), abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR),
            abi.make_op(abi.OP_COARSE_SOLVE, lo)]
    for L in range(lo + 1, hi + 1):
        ops += [abi.make_op(abi.OP_PROLONG_ADD, L, abi.F_SOL, abi.F_SOL, omega=1.0), lost(L), lost(L)]
    monkeypatch.setenv("EVOMG_STRIP3D_MIN_N", "33")
    ho = oracle.create(pd)
    oo = ho.evaluate(ops, 1, 0, 128)
    assert oo.status == abi.SOLVE_CONVERGED and oo.n_iterations > 3      # the correction arrived: the cycle converges
    results = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("EVOMG_NO_FUSE", "1")
        else:
            monkeypatch.delenv("EVOMG_NO_FUSE", raising=False)
        hg = native.create(pd)
        og = hg.evaluate(ops, 1, 0, 128)
        compare_solves(og, oo)
        results.append((og.history, hg.get_field(hi, abi.F_SOL)))
        assert np.array_equal(results[-1][1], ho.get_field(hi, abi.F_SOL))
        hg.close()
    assert np.array_equal(results[0][0], results[1][0])


@pytest.mark.parametrize("smoother", ["jacobi", "block211"])
def test_diverging_candidate_3d_tiles(native, oracle, smoother, monkeypatch):
    """A diverging candidate (over-relaxed Jacobi / (2,1,1) block sweeps, omega = 1.9, textbook semantics) on the 3D tile
    kernels, whose quotients go through const_div (csrc/evomg_scalar.h): the residual norm overflows at the same
    iteration on both sides, both report DIVERGED, and every finite entry of the two histories agrees to 1e-10."""
    from test_full_size_gpu import vcycle
    lo, hi = 4, 6
    pd = problems.poisson_3d(lo, hi)
    block = (2, 1, 1) if smoother == "block211" else (1, 1, 1)
    ops = vcycle(lo, hi, 2, 2, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, block, 1.9))
    monkeypatch.setenv("EVOMG_STRIP3D_MIN_N", "33")
    hg, ho = make_pair(native, oracle, pd)
    og = hg.evaluate(ops, 1, 0, 128)
    oo = ho.evaluate(ops, 1, 0, 128)
    assert oo.status == abi.SOLVE_DIVERGED and og.status == abi.SOLVE_DIVERGED
    assert og.n_iterations == oo.n_iterations and len(og.history) == len(oo.history)
    finite = np.isfinite(oo.history)
    assert np.array_equal(finite, np.isfinite(og.history)) and finite.sum() >= 3
    rel = np.abs(og.history[finite] - oo.history[finite]) / oo.history[finite]
    assert rel.max() <= REL_TOL
    hg.close()


@pytest.mark.parametrize("kind,lo", [("helmholtz", 3), ("helmholtz", 2), ("helmholtz", 1), ("poisson", 3), ("poisson", 2)])
def test_one_wave_bicgstab_is_bit_identical(native, oracle, kind, lo, monkeypatch):
    """BiCGStab on 2D grids up to 9^2 in one wave (evomg_bicgstab_wave2d, lane = 8 * column + row: the wave tree then IS the
    canonical column-tree / row-tree order) against the 512-thread global-memory kernel (EVOMG_NO_REG_KRYLOV) and the
    oracle: solution and residual fields incl. the Robin columns, bit for bit. Complex with the Helmholtz Robin condition
    and a real problem with BiCGStab as its coarse solver."""
    import dataclasses
    if kind == "helmholtz":
        pd = problems.helmholtz_2d(lo, lo + 1, k=0.625 * (1 << (lo + 1)))
    else:
        pd = dataclasses.replace(problems.poisson_2d(lo, lo + 1), cgs_kind=abi.CGS_BICGSTAB, cgs_target_res_reduction=1e-9)
    n = (1 << lo) + 1
    rng = np.random.default_rng(17 + lo)
    cplx = kind == "helmholtz"
    rhs = random_field(rng, (n, n), cplx)
    x0 = zero_boundary(random_field(rng, (n, n), cplx)) * 1e-2
    ops = [abi.make_op(abi.OP_COARSE_SOLVE, lo)]
    ho = oracle.create(pd)
    ho.set_field(lo, abi.F_CGS_RHS, rhs)
    ho.set_field(lo, abi.F_CORR, x0)
    ho.run_ops(ops)
    want = ho.get_field(lo, abi.F_CORR), ho.get_field(lo, abi.F_RES)
    for knob in (None, "EVOMG_NO_REG_KRYLOV"):
        if knob:
            monkeypatch.setenv(knob, "1")
        h = native.create(pd)
        h.set_field(lo, abi.F_CGS_RHS, rhs)
        h.set_field(lo, abi.F_CORR, x0)
        h.run_ops(ops)
        got = h.get_field(lo, abi.F_CORR), h.get_field(lo, abi.F_RES)
        h.close()
        if knob:
            monkeypatch.delenv(knob)
        assert np.array_equal(got[0], want[0]), (kind, lo, knob)
        assert np.array_equal(got[1][1:-1, :], want[1][1:-1, :]), (kind, lo, knob)
    assert not np.array_equal(want[0], x0)


@pytest.mark.parametrize("kind,lo", [("helmholtz", 7), ("helmholtz", 8), ("poisson", 7)])
def test_large_coarsest_grids_run_bicgstab_as_launches(native, oracle, kind, lo, monkeypatch):
    """129^2 and up (C3's coarsest grid): BiCGStab as many-unit launches with the scalars in a device-side state — three per
    iteration (kb2_ps / kb2_hst / kb_xr: p and s evaluated at the nodes the stencil reads, the convergence test at the head of
    the next iteration) or round 4's six (EVOMG_NO_KB_FUSE) —, Robin columns written beside their inner neighbours, dot products
    in the blocked order with fused accumulation that the oracle restates: solution and residual incl. the Robin columns bit
    for bit in both forms (caps of 7, 3, 2 and 1 iterations exercise the batches 2 + 4 + 1, a batch cut short and kernels
    launched past the last iteration; a zero right-hand side the solve that is done before it iterates)."""
    import dataclasses
    if kind == "helmholtz":
        pd = problems.helmholtz_2d(lo, lo + 1, k=0.625 * (1 << (lo + 1)))
    else:
        pd = dataclasses.replace(problems.poisson_2d(lo, lo + 1), cgs_kind=abi.CGS_BICGSTAB, cgs_target_res_reduction=1e-9)
    n = (1 << lo) + 1
    cplx = kind == "helmholtz"
    for cap in (None, 7, 3, 2, 1, "zero"):
        pdc = pd if cap in (None, "zero") else dataclasses.replace(pd, cgs_max_iterations=cap)
        rng = np.random.default_rng(29 + lo)
        rhs = random_field(rng, (n, n), cplx)
        x0 = zero_boundary(random_field(rng, (n, n), cplx)) * 1e-2
        if cap == "zero":
            rhs, x0 = rhs * 0, x0 * 0
        ops = [abi.make_op(abi.OP_COARSE_SOLVE, lo)]
        ho = oracle.create(pdc)
        ho.set_field(lo, abi.F_CGS_RHS, rhs)
        ho.set_field(lo, abi.F_CORR, x0)
        ho.run_ops(ops)
        want = ho.get_field(lo, abi.F_CORR), ho.get_field(lo, abi.F_RES)
        ho.close()
        for knob in (None, "EVOMG_NO_KB_FUSE"):
            if knob:
                monkeypatch.setenv(knob, "1")
            h = native.create(pdc)
            h.set_field(lo, abi.F_CGS_RHS, rhs)
            h.set_field(lo, abi.F_CORR, x0)
            h.run_ops(ops)
            got = h.get_field(lo, abi.F_CORR), h.get_field(lo, abi.F_RES)
            h.run_ops(ops)                                   # a second solve on the same handle: the buffers of p and nu start over
            again = h.get_field(lo, abi.F_CORR)
            h.close()
            if knob:
                monkeypatch.delenv(knob)
            assert np.array_equal(got[0], want[0]), (kind, lo, cap, knob)
            assert np.array_equal(got[1][1:-1, :], want[1][1:-1, :]), (kind, lo, cap, knob)
            assert np.all(np.isfinite(again)), (kind, lo, cap, knob)
        if cap != "zero":
            assert not np.array_equal(want[0], x0)


@pytest.mark.parametrize("kind,smoother,pre,post", [("poisson", "rbgs", 1, 1), ("poisson", "jacobi", 2, 1), ("helmholtz", "rbgs", 2, 1),
                                                    ("helmholtz", "jacobi", 1, 2), ("poisson", "rbgs", 0, 2)])
def test_small_grid_lds_sweeps_do_not_change_a_bit(native, kind, smoother, pre, post, monkeypatch):
    """Point smoothers with the field, or independent tiles of it with a recomputed halo, in LDS (evomg_small2d: plain,
    zero-input, prolongation + sweep, sweep + residual + restriction, the Robin columns of Helmholtz problems at every
    stage) against the marching-strip kernels they replace (EVOMG_NO_SMALL2D): identical residual histories, solutions,
    residuals and coarse right-hand sides on every level of a 9^2 .. 129^2 (Helmholtz: 257^2) hierarchy."""
    from test_full_size_gpu import vcycle
    lo, hi = (3, 7) if kind == "poisson" else (3, 8)       # Helmholtz: 257^2 as 8 x 8 tiles as well (its default since round 5)
    pd = problems.poisson_2d(lo, hi) if kind == "poisson" else problems.helmholtz_2d(lo, hi, k=0.625 * (1 << hi))
    part, om = (abi.PART_RED_BLACK, 1.1 if kind == "poisson" else 0.6) if smoother == "rbgs" else (abi.PART_SINGLE, 0.8 if kind == "poisson" else 0.6)
    ops = vcycle(lo, hi, pre, post, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, part, (1, 1, 1), om))
    monkeypatch.setenv("EVOMG_SMALL2D_MAX_N", "129" if kind == "poisson" else "257")      # 65^2, 129^2 (257^2) as 2 x 2, 4 x 4 (8 x 8) independent tiles with halos
    results = []
    for knob in (None, "EVOMG_NO_SMALL2D"):
        if knob:
            monkeypatch.setenv(knob, "1")
        h = native.create(pd)
        out = h.evaluate(ops, 1, 0, 64)
        results.append((out, [h.get_field(L, f) for L in range(lo + 1, hi + 1) for f in (abi.F_SOL, abi.F_RHS)] + [h.get_field(hi, abi.F_RES)]))
        h.close()
        if knob:
            monkeypatch.delenv(knob)
    (a, fa), (b, fb) = results
    assert a.n_iterations == b.n_iterations and a.n_iterations >= 1
    assert np.array_equal(a.history, b.history)
    for x, y in zip(fa, fb):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("smoother,hi", [("rbgs", 7), ("jacobi", 7), ("rbgs", 9)])
def test_complex_relaxations_divide_by_the_centre_with_a_hoisted_reciprocal(native, oracle, smoother, hi, monkeypatch):
    """A complex relaxation divides by the stencil's centre: two real quotients by |c0|^2, the same divisor at every node of a
    launch. The kernels (strips and LDS tiles) run the compiler's own division sequence with its divisor-only part formed once
    per launch (evomg_strip.h: div_c0, evomg_scalar.h: const_div) — against the plain divisions (EVOMG_NO_CPLX_CONST_DIV):
    identical histories and fields on every level; and the oracle's history to 1e-10 on the small hierarchy."""
    from test_full_size_gpu import vcycle
    lo = 3
    pd = problems.helmholtz_2d(lo, hi, k=0.625 * (1 << hi))
    part, om = (abi.PART_RED_BLACK, 0.6) if smoother == "rbgs" else (abi.PART_SINGLE, 0.6)
    ops = vcycle(lo, hi, 2, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, part, (1, 1, 1), om))
    results = []
    for knob in (None, "EVOMG_NO_CPLX_CONST_DIV"):
        if knob:
            monkeypatch.setenv(knob, "1")
        h = native.create(pd)
        out = h.evaluate(ops, 1, 0, 64)
        results.append((out, [h.get_field(L, f) for L in range(lo + 1, hi + 1) for f in (abi.F_SOL, abi.F_RHS)] + [h.get_field(hi, abi.F_RES)]))
        h.close()
        if knob:
            monkeypatch.delenv(knob)
    (a, fa), (b, fb) = results
    assert a.n_iterations == b.n_iterations and a.n_iterations >= 1
    assert np.array_equal(a.history, b.history)
    for x, y in zip(fa, fb):
        assert np.array_equal(x, y)
    if hi <= 7:
        oo = oracle.create(pd).evaluate(ops, 1, 0, 64)
        assert oo.n_iterations == a.n_iterations
        assert (np.abs(a.history - oo.history) / oo.history).max() <= REL_TOL


@pytest.mark.parametrize("kind,lo", [("helmholtz", 3), ("helmholtz", 2), ("poisson", 3)])
def test_one_wave_bicgstab_folds_its_copy_and_zero(native, oracle, kind, lo, monkeypatch):
    """`gen_rhs = RHS; gen_error = 0; gen_mgCycle@min()` on grids up to 9^2 with BiCGStab as ONE launch (the kernel copies the
    right-hand side, starts from x = 0 without reading x and leaves the Robin columns as the SET_ZERO's boundary pass and the
    solver would have) — against the three separate ops, the unfused program and the oracle, bit for bit incl. the Robin
    columns; a zero right-hand side (no iteration at all); a non-zero boundary on x (no fold)."""
    import dataclasses
    if kind == "helmholtz":
        pd = problems.helmholtz_2d(lo, lo + 1, k=0.625 * (1 << (lo + 1)))
    else:
        pd = dataclasses.replace(problems.poisson_2d(lo, lo + 1), cgs_kind=abi.CGS_BICGSTAB, cgs_target_res_reduction=1e-9)
    n = (1 << lo) + 1
    cplx = kind == "helmholtz"
    rng = np.random.default_rng(70 + lo)
    copy = abi.make_op(abi.OP_COPY, lo, abi.F_CGS_RHS, abi.F_RHS)
    zero = abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR)
    solve = abi.make_op(abi.OP_COARSE_SOLVE, lo)
    for rhs, x0 in ((random_field(rng, (n, n), cplx), None), (np.zeros((n, n), dtype=complex if cplx else float), None),
                    (random_field(rng, (n, n), cplx), random_field(rng, (n, n), cplx))):
        ho = oracle.create(pd)
        ho.set_field(lo, abi.F_RHS, rhs)
        if x0 is not None:
            ho.set_field(lo, abi.F_SOL, x0)
        ho.run_ops([copy, zero, solve])
        want = [ho.get_field(lo, f) for f in (abi.F_SOL, abi.F_RES, abi.F_CGS_RHS)]
        for programs, knob in (([[copy, zero, solve]], None), ([[copy], [zero], [solve]], None), ([[copy, zero, solve]], "EVOMG_NO_FUSE")):
            if knob:
                monkeypatch.setenv(knob, "1")
            h = native.create(pd)
            h.set_field(lo, abi.F_RHS, rhs)
            if x0 is not None:
                h.set_field(lo, abi.F_SOL, x0)
            for prog in programs:
                h.run_ops(prog)
            got = [h.get_field(lo, f) for f in (abi.F_SOL, abi.F_RES, abi.F_CGS_RHS)]
            h.close()
            if knob:
                monkeypatch.delenv(knob)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]), (kind, programs, knob, x0 is None)
            assert np.array_equal(got[1][1:-1, :], want[1][1:-1, :]), (kind, programs, knob)


@pytest.mark.parametrize("pre,post", [(2, 1), (1, 2), (3, 0)])
def test_small_3d_level_visits_do_not_change_a_bit(native, pre, post, monkeypatch):
    """Levels up to 17^3: a whole visit of the level — [u = 0 | u += w P coarse], the red-black sweeps, residual and
    restriction — as ONE launch with the field in LDS (evomg_small3d_chain) against one launch per op and colour
    (EVOMG_NO_SMALL3D): identical residual histories and fields on every level of a 5^3 .. 33^3 hierarchy."""
    from test_full_size_gpu import vcycle
    lo, hi = 2, 5
    pd = problems.poisson_3d(lo, hi)
    ops = vcycle(lo, hi, pre, post, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.25))
    results = []
    for knob in (None, "EVOMG_NO_SMALL3D"):
        if knob:
            monkeypatch.setenv(knob, "1")
        h = native.create(pd)
        out = h.evaluate(ops, 1, 0, 64)
        results.append((out, [h.get_field(L, f) for L in range(lo + 1, hi + 1) for f in (abi.F_SOL, abi.F_RHS, abi.F_RES)]))
        h.close()
        if knob:
            monkeypatch.delenv(knob)
    (a, fa), (b, fb) = results
    assert a.n_iterations == b.n_iterations and a.n_iterations >= 1
    assert np.array_equal(a.history, b.history)
    for x, y in zip(fa, fb):
        assert np.array_equal(x, y)
