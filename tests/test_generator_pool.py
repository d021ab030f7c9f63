"""ProgramGeneratorPool: the thread-safe facade that keeps several candidates in flight behind the ProgramGenerator
interface (host logic on the CPU; the GPU test compares pooled against sequential evaluation)."""
import threading
from multiprocessing.pool import ThreadPool

import numpy as np
import pytest

from evostencils_amd.code_generation.hip import ProgramGeneratorPool
from evostencils_amd.code_generation import lowering


class _FakeGenerator:
    """Records what reaches it; `evaluate_op_program` blocks until two calls are inside (proves concurrency)."""
    barrier = None

    def __init__(self):
        self._cycle_functions = {}
        self._counter = 7
        self._average_generation_time = 3
        self.dimension = 2
        self.calls, self.inits = [], []
        self.lowering_now = 0
        self.closed = False

    def initialize_code_generation(self, lo, hi):
        self.inits.append((lo, hi))

    def reinitialize(self, lo, hi, g):
        self.inits.append(("re", lo, hi, dict(g)))

    def generate_op_program(self, expression, lo, hi, solver_program=""):
        if expression == "bad":
            raise lowering.UnsupportedExpression("bad")
        return ["op", expression]

    def evaluate_op_program(self, ops, lo, hi, solver_program, infinity, samples, gv):
        self.calls.append(ops[1])
        if _FakeGenerator.barrier is not None:
            _FakeGenerator.barrier.wait(timeout=30)
        return (1.0 + ops[1], 0.5, 9.0)

    def generate_cycle_function(self, *a, **k):
        self._cycle_functions["text"] = (1, 0, [])
        return "text"

    def close(self):
        self.closed = True


def test_pool_dispatches_concurrently_and_forwards_the_interface():
    pool = ProgramGeneratorPool(_FakeGenerator, 3)
    gens = pool.generators
    assert pool.dimension == 2 and pool._counter == 7
    pool._counter = 0                                   # the optimizer's reset reaches every worker
    pool._average_generation_time = 0
    assert all(g._counter == 0 and g._average_generation_time == 0 for g in gens)
    pool.initialize_code_generation(5, 9)
    pool.reinitialize(6, 10, {"k": 160.0})
    assert all(g.inits == [(5, 9), ("re", 6, 10, {"k": 160.0})] for g in gens)
    assert pool.generate_cycle_function(None) == "text"
    assert all(g._cycle_functions is gens[0]._cycle_functions for g in gens)      # one table of earlier runs' functions
    _FakeGenerator.barrier = threading.Barrier(3)       # three evaluations must be inside at the same time
    try:
        with ThreadPool(3) as tp:
            out = tp.map(lambda i: pool.generate_and_evaluate(i, None, 5, 9, ""), range(3))
    finally:
        _FakeGenerator.barrier = None
    assert sorted(o[0] for o in out) == [1.0, 2.0, 3.0]
    assert sorted(c for g in gens for c in g.calls) == [0, 1, 2] and all(len(g.calls) == 1 for g in gens)
    assert pool.generate_and_evaluate("bad", None, 5, 9, "", infinity=1e100) == (1e100, 1e100, 1e100)
    pool.close()
    assert all(g.closed for g in gens)
    with pytest.raises(ValueError):
        ProgramGeneratorPool(_FakeGenerator, 0)


@pytest.mark.gpu
def test_pooled_evaluation_equals_sequential_on_the_gpu():
    """Twelve seeded candidates (2D Poisson 257^2, four levels) through a pool of four workers from four threads: the same
    convergence factors and iteration counts as one generator evaluating them one after the other."""
    from evostencils_amd import candidates, native, problems
    from evostencils_amd.code_generation.hip import ProgramGenerator
    from helpers import context_for
    lo, hi = 5, 8

    def make():
        g = ProgramGenerator(problem=problems.poisson_2d(lo, hi), library=native.library())
        g.initialize_code_generation(lo, hi)
        return g
    single = make()
    pop = candidates.population(lambda: context_for(single, lo, hi), hi - lo, 2, 12, seed=7)
    want = [single.generate_and_evaluate(e, None, lo, hi, "", evaluation_samples=1) for e in pop]
    single.close()
    pool = ProgramGeneratorPool(make, 4)
    with ThreadPool(4) as tp:
        got = tp.map(lambda e: pool.generate_and_evaluate(e, None, lo, hi, "", evaluation_samples=1), pop)
    pool.close()
    assert [g[1:] for g in got] == [w[1:] for w in want]
    assert all(np.isfinite(g[0]) and g[0] > 0 for g in got)


def test_isolated_time_objective_probes_run_alone():
    """time_objective="isolated": a probe never overlaps a solve or another probe, every evaluation returns the probe's time
    instead of its contended wall time, rho and iterations are untouched, failed candidates are not probed."""
    state = {"solving": 0, "probing": 0, "overlap": False, "probes": 0}
    lock = threading.Lock()

    class Gen(_FakeGenerator):
        def evaluate_op_program(self, ops, lo, hi, solver_program, infinity, samples, gv):
            import time
            with lock:
                state["solving"] += 1
                state["overlap"] |= state["probing"] > 0
            time.sleep(0.01 * (1 + ops[1] % 3))
            with lock:
                state["solving"] -= 1
            if ops[1] == 5:
                return (infinity, infinity, infinity)
            return (1000.0 + ops[1], 0.5, 9.0)

        def isolated_time_ms(self, ops, lo, hi, iterations, solver_program="", gv=None, coarse_iterations=None, probe_iterations=3):
            import time
            with lock:
                state["probing"] += 1
                state["probes"] += 1
                state["overlap"] |= state["solving"] > 0 or state["probing"] > 1
            time.sleep(0.002)
            with lock:
                state["probing"] -= 1
            return 7.0 + ops[1]

    pool = ProgramGeneratorPool(Gen, 4, time_objective="isolated")
    with ThreadPool(4) as tp:
        out = tp.map(lambda i: pool.generate_and_evaluate(i, None, 5, 9, "", infinity=1e100), range(14))
    pool.close()
    assert not state["overlap"] and state["probes"] == 13
    for i, (t, rho, its) in enumerate(out):
        if i == 5:
            assert (t, rho, its) == (1e100, 1e100, 1e100)
        else:
            assert (t, rho, its) == (7.0 + i, 0.5, 9.0)
    with pytest.raises(ValueError):
        ProgramGeneratorPool(Gen, 2, time_objective="fastest")


def test_a_solve_that_raises_last_still_hands_the_probes_over():
    """time_objective="isolated": worker A finishes its solve and parks behind its probe request while worker B is still
    solving; B's solve raises (a device error, a missed coarse-grid rendezvous). B was the last solve in flight, so B has to run
    A's probe on its way out — otherwise A waits for ever and the pool hangs instead of failing."""
    a_parked = threading.Event()

    class Gen(_FakeGenerator):
        def evaluate_op_program(self, ops, lo, hi, solver_program, infinity, samples, gv):
            if ops[1] == "raises":
                assert a_parked.wait(timeout=30)          # A's request is queued (or about to be: _leave sorts that out)
                import time
                time.sleep(0.05)
                raise RuntimeError("device error")
            return (1000.0, 0.5, 9.0)

        def isolated_time_ms(self, *args, **kw):
            return 7.0

    pool = ProgramGeneratorPool(Gen, 2, time_objective="isolated")
    results = {}

    def run(name):
        try:
            if name == "parks":
                threading.Timer(0.02, a_parked.set).start()
            results[name] = pool.generate_and_evaluate(name, None, 5, 9, "", infinity=1e100)
        except RuntimeError as exc:
            results[name] = exc

    threads = [threading.Thread(target=run, args=(name,), daemon=True) for name in ("raises", "parks")]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=20)
    assert not any(t.is_alive() for t in threads), "a worker is still parked behind its probe request"
    assert isinstance(results["raises"], RuntimeError)
    assert results["parks"] == (7.0, 0.5, 9.0)
    pool.close()


def test_more_workers_than_the_hardware_queues_carry_is_said_aloud():
    """More than sixteen handles per process slow every launch of it (profiles/r05_throughput.md): the pool and the evaluator
    warn instead of silently handing the optimizer a distorted time objective."""
    import warnings
    from evostencils_amd import population
    with pytest.warns(RuntimeWarning, match="in flight per GPU"):
        ProgramGeneratorPool(_FakeGenerator, population.MAX_WORKERS_PER_GPU + 1)
    with pytest.warns(RuntimeWarning, match="in flight per GPU"):
        population.ConcurrentEvaluator(_FakeGenerator, population.MAX_WORKERS_PER_GPU + 2)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ProgramGeneratorPool(_FakeGenerator, population.MAX_WORKERS_PER_GPU)
