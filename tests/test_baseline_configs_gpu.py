"""BASELINE.json's configurations at THEIR sizes where earlier suites only had reduced ones:

* C5 — a 16-candidate slice of the seeded `default_rng(2024)` population on 2D Poisson 2049^2, levels 11..7, through
  `population.evaluate_population` and `population.evaluate_population_in_flight` (the reference's per-generation
  evaluation, optimization/program.py:534-536,582): four of them against the oracle, all of them in flight == sequential;
* C3 — the frozen EVOLVED Helmholtz double cycle at 2049^2 (capped), by the invariants of the returned iterate;
* C1 and C4 — the two configs that name a Jacobi-type smoother, under the drop-in's DEFAULT semantics
  (`exastencils-v1.1`: a `with jacobi` sweep never takes effect, include/evomg.h) as well as the textbook one: against the
  oracle where it finishes in seconds, and at C4's full size by the property the semantics implies — the solve is bit for
  bit the solve of the same cycle with its smoothers removed.
"""
import dataclasses

import numpy as np
import pytest

from evostencils_amd import abi, candidates, population, problems
from helpers import context_for
from test_full_size_gpu import vcycle

pytestmark = pytest.mark.gpu

V11 = abi.JACOBI_SEMANTICS["exastencils-v1.1"]
TEXTBOOK = abi.JACOBI_SEMANTICS["textbook"]


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()
    return nat


# ---- C5 ---------------------------------------------------------------------------------------------------------------
C5_LO, C5_HI, C5_SLICE = 7, 11, 16
C5_VS_ORACLE = (0, 1, 6, 11)          # also solved by the CPU oracle (15-35 s each): two converge (56 / 60 iterations), two hit the cap


def _c5_generator(native):
    from evostencils_amd.code_generation.hip import ProgramGenerator
    g = ProgramGenerator(problem=problems.poisson_2d(C5_LO, C5_HI), library=native.library())      # default semantics
    g.initialize_code_generation(C5_LO, C5_HI)
    return g


def test_c5_population_slice_at_full_size(native):
    """configs[4] at its size: the first 16 candidates of the bench's population (same seed, same generator, so the same
    DAGs as `bench.py`'s first 16). Sequential pass == in-flight pass in rho and iterations for all 16; four of them ==
    the oracle (identical iteration count, rho within 1e-10 relative)."""
    from oracle import oracle
    from evostencils_amd.code_generation.hip import ProgramGenerator
    gen = _c5_generator(native)
    pop = candidates.population(lambda: context_for(gen, C5_LO, C5_HI), C5_HI - C5_LO, 2, C5_SLICE, seed=2024)
    rows = population.evaluate_population(lambda i: gen.generate_and_evaluate(pop[i], None, C5_LO, C5_HI, "", evaluation_samples=1),
                                          C5_SLICE)
    assert [r[0] for r in rows] == list(range(C5_SLICE))
    workers = population.ConcurrentEvaluator(lambda: _c5_generator(native), 8)
    try:
        rows_f = population.evaluate_population_in_flight(
            workers, lambda g, i: g.generate_and_evaluate(pop[i], None, C5_LO, C5_HI, "", evaluation_samples=1), C5_SLICE)
    finally:
        workers.close()
    for a, b in zip(rows, rows_f):
        assert a[0] == b[0] and a[1] == b[1] and a[3] == b[3], (a, b)      # index, rho, iterations: bit for bit
        assert np.isfinite(b[2]) and b[2] > 0
    ref = oracle.create(problems.poisson_2d(C5_LO, C5_HI), jacobi_semantics="exastencils-v1.1")
    for i in C5_VS_ORACLE:
        ops = gen.generate_op_program(pop[i], C5_LO, C5_HI)
        _t, rho_ref, its_ref = ProgramGenerator.interpret_outcome(ref.evaluate(ops, 1, 0, 128), 1e100)
        assert rows[i][3] == its_ref, (i, rows[i], its_ref)
        assert rows[i][1] == pytest.approx(rho_ref, rel=1e-10), (i, rows[i], rho_ref)
    ref.close()
    gen.close()
    capped = sum(1 for r in rows if r[3] >= gen.problem.max_iterations)
    print(f"C5 slice: {C5_SLICE} candidates at 2049^2, {capped} at the iteration cap, rho {min(r[1] for r in rows):.4f} .. "
          f"{max(r[1] for r in rows):.4f}; sequential {sum(r[2] for r in rows):.0f} ms of solve time")


# ---- C3 with the evolved cycle ------------------------------------------------------------------------------------------
def test_c3_helmholtz_2048_evolved_cycle(native):
    """configs[2] says "evolved cycle": the frozen evolved preconditioner (two V(2,1) corrections per application, built by
    the grammar's productions) at 2049^2, k h = 0.625, capped at 60 outer iterations (no cycle converges inside the
    reference's own cap at this size) — recurrence residual == true residual of the returned iterate,
    Robin / Dirichlet boundary, bit-for-bit repeatability."""
    from test_helmholtz_oracle import evolved_double_cycle_ops
    lo, hi = 7, 11
    k = 0.625 * (1 << hi)
    gen, ops = evolved_double_cycle_ops(lo, hi, k)
    pd = dataclasses.replace(gen.problem, max_iterations=60)
    h = native.create(pd)
    out = h.evaluate(ops, 1, 600e3, 64)
    assert out.status == abi.SOLVE_MAXITER and out.n_iterations == 60
    assert np.all(np.isfinite(out.history)) and len(out.history) == 61
    u = h.get_field(hi, abi.F_CORR)
    n = (1 << hi) + 1
    hh = 1.0 / (n - 1)
    assert np.allclose(u[1:-1, 0], u[1:-1, 1] / (1 - 1j * k * hh), rtol=1e-12, atol=0)
    assert np.allclose(u[1:-1, -1], u[1:-1, -2] / (1 - 1j * k * hh), rtol=1e-12, atol=0)
    assert np.all(u[0, :] == 0) and np.all(u[-1, :] == 0)
    f = pd.right_hand_side(hi)
    Au = (4.0 / hh ** 2 - k ** 2) * u[1:-1, 1:-1] - (u[1:-1, :-2] + u[1:-1, 2:] + u[:-2, 1:-1] + u[2:, 1:-1]) / hh ** 2
    r = f[1:-1, 1:-1] - Au
    assert abs(np.sqrt(abs(np.sum(r * r))) - out.final_res_norm) <= 1e-6 * out.init_res_norm
    again = h.evaluate(ops, 1, 600e3, 64)
    assert np.array_equal(again.history, out.history)
    h.close()
    print(f"C3 evolved: 60 outer iterations in {out.time_ms:.0f} ms ({out.time_ms / 60:.2f} ms per iteration = 4 MG cycles + 5 dots)")


def test_c3_thirty_outer_iterations_at_full_size_against_the_oracle(native):
    """C3 at 2049^2 (complex, k h = 0.625, the evolved double cycle as preconditioner) for the thirty outer BiCGStab iterations
    `bench.py`'s `configs.c3` times, against the oracle: every norm of the history within 1e-10 (north_star's tolerance), the iterate within 1e-10 of its scale."""
    from oracle import oracle
    from test_helmholtz_oracle import evolved_double_cycle_ops
    lo, hi = 7, 11
    gen, ops = evolved_double_cycle_ops(lo, hi, 0.625 * (1 << hi))
    pd = dataclasses.replace(gen.problem, max_iterations=30)
    hg = native.create(pd)
    a = hg.evaluate(ops, 1, 600e3, 40)
    ug = hg.get_field(hi, abi.F_CORR)
    hg.close()
    ho = oracle.create(pd)
    b = ho.evaluate(ops, 1, 600e3, 40)
    uo = ho.get_field(hi, abi.F_CORR)
    ho.close()
    assert a.n_iterations == b.n_iterations == 30 and len(a.history) == len(b.history) == 31
    assert np.max(np.abs(a.history - b.history) / np.abs(b.history)) <= 1e-10
    assert np.max(np.abs(ug - uo)) <= 1e-10 * np.max(np.abs(uo))


# ---- C1 / C4 in both `with jacobi` semantics ------------------------------------------------------------------------------
def _jacobi(L, omega=0.8):
    return abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1, 1, 1), omega)


def _block211(L, omega=0.8):
    return abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 1, 1), omega)


def _without_smoothers(ops):
    return [op for op in ops if op.opcode != abi.OP_SMOOTH]


@pytest.mark.parametrize("semantics", ["textbook", "exastencils-v1.1"])
def test_c1_v11_jacobi_both_semantics_against_the_oracle(native, semantics):
    """configs[0]: 2D Poisson 256^2, 3-level V(1,1) Jacobi w = 0.8. Textbook: converges (rho ~ 0.34); reference pipeline's
    semantics: the Jacobi sweeps never take effect, the cycle is a bare coarse-grid correction and runs into the
    100-iteration cap — exactly what the oracle does."""
    from oracle import oracle
    lo, hi = 6, 8
    pd = problems.poisson_2d(lo, hi)
    ops = vcycle(lo, hi, 1, 1, _jacobi)
    hg = native.create(pd, jacobi_semantics=semantics)
    ho = oracle.create(pd, jacobi_semantics=semantics)
    a, b = hg.evaluate(ops, 1, 0, 128), ho.evaluate(ops, 1, 0, 128)
    assert a.status == b.status and a.n_iterations == b.n_iterations
    assert np.all(np.abs(a.history - b.history) <= 1e-10 * b.history)
    assert np.array_equal(hg.get_field(hi, abi.F_SOL), ho.get_field(hi, abi.F_SOL))
    if semantics == "textbook":
        assert a.status == abi.SOLVE_CONVERGED and a.conv_factor < 0.5
    else:
        assert a.status == abi.SOLVE_MAXITER and a.n_iterations == pd.max_iterations
        bare = hg.evaluate(_without_smoothers(ops), 1, 0, 128)
        assert np.array_equal(bare.history, a.history)
    print(f"C1 [{semantics}]: status {a.status}, {a.n_iterations:.0f} iterations, rho {a.conv_factor:.6f}")
    hg.close()
    ho.close()


@pytest.mark.parametrize("semantics", ["textbook", "exastencils-v1.1"])
def test_c4_cycle_at_129_both_semantics_against_the_oracle(native, semantics):
    """C4's cycle (V(2,2) block-Jacobi (2,1,1), w = 0.8) on 129^3, the size the oracle finishes in seconds."""
    from oracle import oracle
    lo, hi = 5, 7
    pd = dataclasses.replace(problems.poisson_3d(lo, hi), max_iterations=20)
    ops = vcycle(lo, hi, 2, 2, _block211)
    hg = native.create(pd, jacobi_semantics=semantics)
    ho = oracle.create(pd, jacobi_semantics=semantics)
    a, b = hg.evaluate(ops, 1, 0, 128), ho.evaluate(ops, 1, 0, 128)
    assert a.status == b.status and a.n_iterations == b.n_iterations
    assert np.all(np.abs(a.history - b.history) <= 1e-10 * b.history)
    assert np.array_equal(hg.get_field(hi, abi.F_SOL), ho.get_field(hi, abi.F_SOL))
    if semantics == "exastencils-v1.1":
        assert a.status != abi.SOLVE_CONVERGED
    hg.close()
    ho.close()


@pytest.mark.parametrize("smoother", ["block211", "jacobi"])
def test_c4_launch_plan_at_129_against_the_oracle(native, monkeypatch, smoother):
    """C4's launch plan — double sweeps in one pass, the prolongation folded into the post-smoothing pair, residual +
    restriction in one pass, and the loop's residual norm riding on the COMING cycle's first double sweep (round 4:
    evomg_tile3d_twice<NORM0>, launched speculatively) — runs from 257^3 by default; forced onto 129^3 here, where the oracle
    answers in seconds. Histories bit-identical to the oracle's and to the plan with a norm pass of its own."""
    from oracle import oracle
    monkeypatch.setenv("EVOMG_STRIP3D_MIN_N", "129")
    lo, hi = 5, 7
    pd = dataclasses.replace(problems.poisson_3d(lo, hi), max_iterations=20)
    ops = vcycle(lo, hi, 2, 2, _block211 if smoother == "block211" else _jacobi)
    hg = native.create(pd, jacobi_semantics="textbook")
    ho = oracle.create(pd, jacobi_semantics="textbook")
    # the plan is the one meant: the loop norm is timed as ONE pass that also writes the swept field (3 fields, not 2)
    hg.profile_begin(abi.PROF_LOOP_NORM, hi)
    a = hg.evaluate(ops, 1, 0, 128)
    prof = hg.profile_end()
    n = (1 << hi) + 1
    # (one norm per iteration + the initial one; the norm after the last iteration the loop may run has no coming cycle)
    fields = prof.bytes * prof.launches / (8.0 * n ** 3)
    assert prof.launches == a.n_iterations + 1 and abs(fields - (3.0 * a.n_iterations + 2.0)) < 1e-6
    b = ho.evaluate(ops, 1, 0, 128)
    assert a.status == b.status and a.n_iterations == b.n_iterations
    assert np.array_equal(a.history, b.history)
    assert np.array_equal(hg.get_field(hi, abi.F_SOL), ho.get_field(hi, abi.F_SOL))
    assert np.array_equal(hg.get_field(hi, abi.F_RES), ho.get_field(hi, abi.F_RES))
    hg.close()
    monkeypatch.setenv("EVOMG_NO_HEAD_NORM", "1")
    hp = native.create(pd, jacobi_semantics="textbook")
    c = hp.evaluate(ops, 1, 0, 128)
    assert np.array_equal(a.history, c.history) and a.n_iterations == c.n_iterations
    assert np.array_equal(hp.get_field(hi, abi.F_SOL), ho.get_field(hi, abi.F_SOL))
    hp.close()
    ho.close()


GEOMETRIES = {
    "balanced": {},
    "depth_12": {"EVOMG_STRIP_PLANES": "12"},
    "depth_44": {"EVOMG_STRIP_PLANES": "44"},
    "depth_capped_20": {"EVOMG_BALANCE3D": "20"},
    "fixed_16_planes": {"EVOMG_NO_BALANCE3D": "1"},
}


@pytest.mark.parametrize("geometry", sorted(GEOMETRIES))
@pytest.mark.parametrize("smoother", ["rbgs", "block211"])
def test_3d_tile_geometries_are_bit_identical_to_the_oracle(native, monkeypatch, smoother, geometry):
    """The launch geometry of the 3D tiles (round 4: z-chunk depth picked by workgroup balance, strips of equal width) must
    not show in a single bit: the eight-wave tiles 513^3 runs on, forced onto 129^3 with several chunk depths, against the
    oracle — V(2,1) cycles, so that single sweeps, double sweeps, fused residual + restriction, the folded prolongation and
    the loop norm all take part."""
    from oracle import oracle
    monkeypatch.setenv("EVOMG_STRIP3D_MIN_N", "65")
    monkeypatch.setenv("EVOMG_TILE3D_WAVES", "8")
    for k, v in GEOMETRIES[geometry].items():
        monkeypatch.setenv(k, v)
    lo, hi = 5, 7
    pd = dataclasses.replace(problems.poisson_3d(lo, hi), max_iterations=6)
    rb = lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.15)
    ops = vcycle(lo, hi, 2, 1, rb if smoother == "rbgs" else _block211)
    hg = native.create(pd, jacobi_semantics="textbook")
    ho = oracle.create(pd, jacobi_semantics="textbook")
    a, b = hg.evaluate(ops, 1, 0, 16), ho.evaluate(ops, 1, 0, 16)
    assert a.n_iterations == b.n_iterations and np.array_equal(a.history, b.history)
    assert np.array_equal(hg.get_field(hi, abi.F_SOL), ho.get_field(hi, abi.F_SOL))
    hg.close()
    ho.close()


@pytest.mark.parametrize("top,iterations,smoother", [(8, 4, "block211"), (9, 3, "block211"), (9, 1, "rbgs"), (9, 1, "jacobi")])
def test_c4_first_cycles_at_full_size_against_the_oracle(native, top, iterations, smoother):
    """C4 itself (textbook semantics, default thresholds and launch geometry: nothing forced) for its first cycle(s) against the
    oracle — at 257^3 (two cycles) and at 513^3 (one cycle: the 172-plane chunks, the 104-column strips, the norm riding on
    the first double sweep, the 33^3 CG with its exchange words): norms and the whole solution field bit-identical. One cycle
    each with the fused red-black tiles (V(2,1)) and point Jacobi in C4's place."""
    from oracle import oracle
    lo = 5
    pd = dataclasses.replace(problems.poisson_3d(lo, top), max_iterations=iterations)
    if smoother == "rbgs":
        ops = vcycle(lo, top, 2, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.15))
    else:
        ops = vcycle(lo, top, 2, 2, _block211 if smoother == "block211" else _jacobi)
    hg = native.create(pd, jacobi_semantics="textbook")
    a = hg.evaluate(ops, 1, 0, 8)
    ug = hg.get_field(top, abi.F_SOL)
    hg.close()
    ho = oracle.create(pd, jacobi_semantics="textbook")
    b = ho.evaluate(ops, 1, 0, 8)
    assert a.n_iterations == b.n_iterations == iterations
    assert np.array_equal(a.history, b.history)
    assert np.array_equal(ug, ho.get_field(top, abi.F_SOL))
    ho.close()


def test_c4_full_size_in_the_reference_pipelines_semantics(native):
    """configs[3] at 513^3 under `exastencils-v1.1`: every block-Jacobi sweep is launched (and timed) but never advanced,
    so the residual history is bit for bit the history of the same cycle with the smoothers removed; the candidate does not
    converge (what the drop-in's default reports for C4; `textbook` is what test_full_size_gpu.py runs)."""
    lo, hi = 5, 9
    pd = dataclasses.replace(problems.poisson_3d(lo, hi), max_iterations=12)
    ops = vcycle(lo, hi, 2, 2, _block211)
    h = native.create(pd, jacobi_semantics="exastencils-v1.1")
    a = h.evaluate(ops, 1, 0, 64)
    bare = h.evaluate(_without_smoothers(ops), 1, 0, 64)
    assert a.status == bare.status == abi.SOLVE_MAXITER and a.n_iterations == 12
    assert np.array_equal(a.history, bare.history)
    assert a.time_ms > bare.time_ms                     # the sweeps ran and cost their time
    print(f"C4 [exastencils-v1.1]: 12 iterations, rho {a.conv_factor:.4f}, {a.time_ms:.0f} ms with the (lost) sweeps, "
          f"{bare.time_ms:.0f} ms without")
    h.close()


# ---- C2 with the device full ------------------------------------------------------------------------------------------------
def test_c2_in_flight_at_full_size_is_the_sequential_solve(native):
    """configs[1] (the headline candidate: 2D Poisson 4097^2, levels 12..7, V(1,1) red-black, CG on 129^2) with EIGHT evaluations
    in flight — the regime in which the cross-iteration launch takes 256-row chunks on levels that stream from HBM
    (csrc/impl_launch2d.inl: cross_rows; round 6) and the coarse solves are dealt to the XCDs in turn — against the same candidate
    evaluated alone: convergence factor and iteration count bit for bit, from every worker, twice."""
    import bench
    from evostencils_amd.code_generation.hip import ProgramGenerator
    lo, hi = 7, 12

    def make():
        g = ProgramGenerator(problem=problems.poisson_2d(lo, hi), library=native.library())
        g.initialize_code_generation(lo, hi)
        return g

    gen = make()
    candidate = bench.build_candidate(gen, lo, hi)
    alone = gen.generate_and_evaluate(candidate, None, lo, hi, "", evaluation_samples=1)
    gen.close()
    assert 0.0 < alone[1] < 0.2 and alone[2] == 12.0
    workers = population.ConcurrentEvaluator(make, 8)
    try:
        rows = population.evaluate_population_in_flight(
            workers, lambda g, i: g.generate_and_evaluate(candidate, None, lo, hi, "", evaluation_samples=1), 16)
    finally:
        workers.close()
    assert len(rows) == 16
    for r in rows:
        assert r[1] == alone[1] and r[3] == alone[2], (r, alone)
