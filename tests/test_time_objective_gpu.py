"""A contention-free time objective for candidates in flight (the reference's second objective is time per
iteration, optimization/program.py:449-451): evomg_time_iterations probes a candidate alone on the GPU, and the modelled
time = probe x iterations has to rank candidates as the sequential evaluation does and agree with it."""
import numpy as np
import pytest

from evostencils_amd import abi, candidates, population, problems
from helpers import context_for

pytestmark = pytest.mark.gpu
LO, HI, N = 7, 11, 16


def _generator():
    from evostencils_amd import native
    from evostencils_amd.code_generation.hip import ProgramGenerator
    g = ProgramGenerator(problem=problems.poisson_2d(LO, HI), library=native.library())
    g.initialize_code_generation(LO, HI)
    return g


def _spearman(a, b):
    ra, rb = np.argsort(np.argsort(a)).astype(float), np.argsort(np.argsort(b)).astype(float)
    ra -= ra.mean()
    rb -= rb.mean()
    return float((ra * rb).sum() / np.sqrt((ra * ra).sum() * (rb * rb).sum()))


def test_probe_runs_exactly_the_iterations_asked_for():
    from evostencils_amd import native
    from test_full_size_gpu import vcycle
    pd = problems.poisson_2d(5, 9)
    h = native.create(pd)
    ops = vcycle(5, 9, 1, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.0))
    full = h.evaluate(ops, 1, 0, 64)
    ms, done = h.time_iterations(ops, 4)
    assert done == 4 and 0 < ms < 50
    again = h.evaluate(ops, 1, 0, 64)                   # the probe re-targets the stopping rule only for its own run
    assert again.n_iterations == full.n_iterations and np.array_equal(again.history, full.history)
    many, done = h.time_iterations(ops, 40)             # far beyond convergence: still all 40 (no convergence test)
    assert done == 40
    h.close()


def test_modelled_time_ranks_candidates_like_the_sequential_pass():
    """The 16-candidate C5 slice at 2049^2: sequential times vs (in flight) rho / iterations + isolated probes."""
    gen = _generator()
    pop = candidates.population(lambda: context_for(gen, LO, HI), HI - LO, 2, N, seed=2024)
    evaluate = lambda g, i: g.generate_and_evaluate(pop[i], None, LO, HI, "", evaluation_samples=1)
    evaluate(gen, 0)                                    # warm-up
    seq = [evaluate(gen, i) for i in range(N)]
    seq2 = [evaluate(gen, i) for i in range(N)]         # the sequential pass against itself: the noise floor
    workers = population.ConcurrentEvaluator(_generator, 8)
    try:
        workers.map(evaluate, range(8))                 # warm-up of every worker
        contended = population.evaluate_population_in_flight(workers, evaluate, N)
        modelled = population.evaluate_population_in_flight(
            workers, evaluate, N,
            time_fn=lambda g, i, its, k: g.isolated_time_ms(g.generate_op_program(pop[i], LO, HI), LO, HI, its, coarse_iterations=k))
    finally:
        workers.close()
    gen.close()
    t_seq = np.array([s[0] for s in seq])
    t_seq2 = np.array([s[0] for s in seq2])
    t_con = np.array([r[2] for r in contended])
    t_mod = np.array([r[2] for r in modelled])
    assert [r[1] for r in modelled] == [s[1] for s in seq] and [r[3] for r in modelled] == [s[2] for s in seq]
    # the reference time of a candidate: the faster of its two sequential solves (a disturbed solve on a shared machine is
    # slower, never faster)
    rel = np.abs(t_mod - np.minimum(t_seq, t_seq2)) / np.minimum(t_seq, t_seq2)
    print(f"sequential {t_seq.sum():.0f} ms; modelled/sequential: max rel diff {rel.max():.3f}, Spearman {_spearman(t_mod, t_seq):.4f}; "
          f"contended/sequential {np.median(t_con / t_seq):.1f}x (Spearman {_spearman(t_con, t_seq):.4f}); "
          f"sequential vs itself: max rel diff {(np.abs(t_seq2 - t_seq) / t_seq).max():.3f}, Spearman {_spearman(t_seq2, t_seq):.4f}")
    # the sequential pass against itself bounds what any model can reach (near-ties between candidates flip ranks)
    assert _spearman(t_mod, t_seq) >= min(0.98, _spearman(t_seq2, t_seq) - 0.02)
    assert rel.max() <= 0.05
    assert _spearman(t_mod, t_seq) > _spearman(t_con, t_seq)


def test_the_coarse_iteration_cost_is_measured_once_and_then_known():
    """One generator models the whole slice: its first three candidates run the three-probe model, the cost of ONE coarse-solver
    iteration is then settled (it belongs to the problem, not to the candidate) and the others run the two cheap probes only
    (evomg_time_model_known_coarse) — every modelled time within 5 % of the candidate's sequential time."""
    # wall-clock comparisons on a shared machine: a disturbed stretch on either side shows as a deviation that has nothing to do
    # with the model (seen once in the round-6 suite runs: every candidate of the slice off by tens of per cent, the same test
    # within 1-2 % three times in a row on the next box) — the measurement may be repeated twice
    for attempt in range(3):
        gen = _generator()
        pop = candidates.population(lambda: context_for(gen, LO, HI), HI - LO, 2, N, seed=2024)
        evaluate = lambda i: gen.generate_and_evaluate(pop[i], None, LO, HI, "", evaluation_samples=1)
        evaluate(0)
        seq, counts = [], []
        for i in range(N):
            first = evaluate(i)
            again = evaluate(i)                              # the faster of two solves: a disturbed one is slower, never faster
            seq.append(first if first[0] <= again[0] else again)
            counts.append(gen.last_coarse_iterations)
        modelled = []
        for i in range(N):
            settled_before = gen.coarse_iteration_cost_ms(LO, HI) is not None
            assert settled_before == (i >= gen.COARSE_COST_SAMPLES)
            modelled.append(gen.isolated_time_ms(gen.generate_op_program(pop[i], LO, HI), LO, HI, seq[i][2], coarse_iterations=counts[i]))
        per_cg = gen.coarse_iteration_cost_ms(LO, HI)
        gen.close()
        t_seq, t_mod = np.array([s[0] for s in seq]), np.array(modelled)
        rel = np.abs(t_mod - t_seq) / t_seq
        print(f"attempt {attempt + 1}: coarse iteration {1e3 * per_cg:.2f} us; modelled/sequential: max rel diff {rel.max():.3f} (known-cost "
              f"candidates {rel[gen.COARSE_COST_SAMPLES:].max():.3f}), Spearman {_spearman(t_mod, t_seq):.4f}")
        assert 0.5e-3 < per_cg < 10e-3                       # 129^2 CG: about 1.5 us per iteration
        if rel.max() <= 0.05 and _spearman(t_mod, t_seq) >= 0.98:
            break
    assert rel.max() <= 0.05 and _spearman(t_mod, t_seq) >= 0.98


def test_pool_with_the_isolated_time_objective_matches_sequential_times():
    """ProgramGeneratorPool(..., time_objective="isolated") behind a thread pool — what stands behind the optimizer's own `map`
    (INTEGRATION.md section 3): rho and iterations of the sequential evaluation, and its TIMES within 5 % although four
    candidates are in flight."""
    from multiprocessing.pool import ThreadPool
    from evostencils_amd.code_generation.hip import ProgramGeneratorPool
    lo, hi, n = LO, HI, 12             # the C5 slice's size: solves of 50-130 ms, as in the test above

    def make():
        from evostencils_amd import native
        from evostencils_amd.code_generation.hip import ProgramGenerator
        g = ProgramGenerator(problem=problems.poisson_2d(lo, hi), library=native.library())
        g.initialize_code_generation(lo, hi)
        return g
    single = make()
    pop = candidates.population(lambda: context_for(single, lo, hi), hi - lo, 2, n, seed=2024)
    evaluate = lambda g, e: g.generate_and_evaluate(e, None, lo, hi, "", evaluation_samples=1)
    evaluate(single, pop[0])
    want = [evaluate(single, e) for e in pop]
    single.close()
    # wall-clock comparisons on a shared machine: a disturbed solve on either side shows as a deviation that has nothing to do
    # with the model (seen once in ~20 runs of the suite: 8.6 % on one candidate) — the measurement may be repeated twice
    for attempt in range(3):
        pool = ProgramGeneratorPool(make, 4, time_objective="isolated")
        with ThreadPool(4) as tp:
            tp.map(lambda e: evaluate(pool, e), pop[:4])                # warm-up of every worker
            got = tp.map(lambda e: evaluate(pool, e), pop)
        pool.close()
        assert [g[1:] for g in got] == [w[1:] for w in want]
        usable = [(g[0], w[0]) for g, w in zip(got, want) if w[0] < 1e99]
        rel = max(abs(a - b) / b for a, b in usable)
        print(f"pool, isolated time objective (attempt {attempt + 1}): {len(usable)} candidates, max rel deviation from sequential {rel:.3f}")
        if rel <= 0.05:
            break
    assert rel <= 0.05


@pytest.mark.parametrize("kind", ["elasticity", "fas"])
def test_time_model_of_problems_whose_coarse_iterations_are_not_counted(kind):
    """Systems of PDEs and FAS problems do not report coarse-solver iterations (evomg_coarse_iterations is 0), so the time
    model must not cap their coarse solver in its probes — nothing would add the cost back. fixed + per_iteration x
    iterations has to land on the time of a sequential solve."""
    import cycles
    from evostencils_amd import native
    if kind == "elasticity":
        from helpers import generator_for
        lo, hi = 4, 8
        gen = generator_for("2D_FD_LinearElasticity_fromL2", lo, hi, library=native.library())
        rb = [("rbgs", 23)]
        expression, _ = cycles.v_cycle(context_for(gen, lo, hi), hi - lo, [rb * 2] * (hi - lo), [rb] * (hi - lo))
        ops = gen.generate_op_program(expression, lo, hi)
        handle = native.create(gen.problem)
    else:
        from test_fas_lowering import fas_context, fas_generator
        gen = fas_generator(library=native.library(), levels=(5, 9))
        sm = [[cycles._nw(14)] for _ in range(4)]
        ops = gen.generate_op_program(cycles.fas_v_cycle(fas_context(gen, 4), 4, sm, sm)[0])
        handle = native.create(gen.problem_description())
    handle.evaluate(ops, 1, 0, 0)
    solves = sorted((handle.evaluate(ops, 1, 0, 0) for _ in range(3)), key=lambda o: o.time_ms)
    ref = solves[0]
    assert ref.n_iterations >= 6 and handle.coarse_iterations() == 0
    fixed, per_iteration, per_coarse = handle.time_model(ops, 2, 6)
    modelled = max(fixed, 0.0) + per_iteration * ref.n_iterations
    print(f"{kind}: sequential {ref.time_ms:.3f} ms over {ref.n_iterations:.0f} iterations, modelled {modelled:.3f} ms "
          f"(fixed {fixed:.3f}, per iteration {per_iteration:.4f}, per coarse iteration {per_coarse})")
    assert per_coarse == 0.0
    assert abs(modelled - ref.time_ms) <= 0.12 * ref.time_ms
    handle.close()
