"""N>1 path on CPU: world_size-2 `gloo` process groups exercise the sharding and the fitness collectives that run
over RCCL (`nccl` backend) on the GPU node. The evaluate callback is a stand-in (no GPU here); what is tested is
the partitioning, padding and ordering of the exchange."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from evostencils_amd import population

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_fitness(i):
    return (10.0 + i, 0.1 + 0.001 * i, float(5 + i % 3))      # (time_ms, rho, iterations)


def _worker(rank, world, port, n_candidates, queue):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        evaluated = []

        def evaluate(i):
            evaluated.append(i)
            return _fake_fitness(i)

        rows = population.evaluate_population(evaluate, n_candidates)
        gathered = population.gather_fitness([(0.5 + rank, 1.0 + rank, 7.0)], world)
        queue.put((rank, evaluated, rows, gathered))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_candidates", [8, 7, 1])
def test_population_is_sharded_round_robin_and_gathered(n_candidates):
    world = 2
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_candidates, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, evaluated, rows, gathered = queue.get(timeout=120)
        results[rank] = (evaluated, rows, gathered)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = [(i, 0.1 + 0.001 * i, 10.0 + i, float(5 + i % 3)) for i in range(n_candidates)]
    for rank in range(world):
        evaluated, rows, gathered = results[rank]
        assert evaluated == [i for i in range(n_candidates) if i % world == rank]      # program.py:534-535
        assert [r[0] for r in rows] == list(range(n_candidates))
        for got, want in zip(rows, expected):
            assert got[0] == want[0] and all(abs(a - b) < 1e-15 for a, b in zip(got[1:], want[1:]))
    # gather to rank 0 in rank order; other ranks keep their local list
    assert results[0][2] == [(0.5, 1.0, 7.0), (1.5, 2.0, 7.0)]
    assert results[1][2] == [(1.5, 2.0, 7.0)]


def _dynamic_worker(rank, world, port, n_candidates, queue):
    import time
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = []
        for generation in range(2):                    # two generations: the queue key must advance
            def evaluate(i):
                time.sleep(0.05 if rank == 0 else 0.005)    # rank 0 is the slow GPU
                return _fake_fitness(i)
            rows, mine = population.evaluate_population_dynamic(evaluate, n_candidates)
            out.append((rows, mine))
        queue.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_candidates", [12, 1])
def test_dynamic_work_queue_hands_every_candidate_out_once(n_candidates):
    world = 2
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dynamic_worker, args=(r, world, port, n_candidates, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, out = queue.get(timeout=120)
        results[rank] = out
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = [(i, 0.1 + 0.001 * i, 10.0 + i, float(5 + i % 3)) for i in range(n_candidates)]
    for generation in range(2):
        taken = sorted(results[0][generation][1] + results[1][generation][1])
        assert taken == list(range(n_candidates))                    # each candidate exactly once
        for rank in range(world):
            assert results[rank][generation][0] == expected          # exact: the other rank adds zeros
        if n_candidates >= 12:
            assert len(results[1][generation][1]) > len(results[0][generation][1])   # the fast rank drew more


def test_single_process_is_identity():
    assert population.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert population.gather_fitness([(0.1, 2.0, 3.0)], 1) == [(0.1, 2.0, 3.0)]
    rows = population.evaluate_population(_fake_fitness, 3)
    assert [r[0] for r in rows] == [0, 1, 2]
    rows, mine = population.evaluate_population_dynamic(_fake_fitness, 3)
    assert [r[0] for r in rows] == [0, 1, 2] and mine == [0, 1, 2]


class _FakeGenerator:
    """Stand-in for a ProgramGenerator worker of the ConcurrentEvaluator: counts what it evaluated."""

    def __init__(self):
        self.seen = []
        self.closed = False

    def close(self):
        self.closed = True


def _in_flight_worker(rank, world, port, n_candidates, queue):
    import time
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        workers = population.ConcurrentEvaluator(_FakeGenerator, 3)

        def fn(generator, i):
            time.sleep(0.01)
            generator.seen.append(i)
            if i == 5 and rank == 1:
                raise RuntimeError("device lost")         # one broken candidate must not block the collective
            return _fake_fitness(i)

        error = None
        try:
            rows = population.evaluate_population_in_flight(workers, fn, n_candidates)
        except RuntimeError as exc:
            error, rows = str(exc), None
        seen = sorted(i for g in workers.generators for i in g.seen)
        used = sum(1 for g in workers.generators if g.seen)
        workers.close()
        queue.put((rank, rows, seen, used, error, all(g.closed for g in workers.generators)))
    finally:
        dist.destroy_process_group()


def test_in_flight_evaluation_keeps_the_split_and_the_collective():
    world, n_candidates = 2, 12
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_in_flight_worker, args=(r, world, port, n_candidates, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, rows, seen, used, error, closed = queue.get(timeout=120)
        results[rank] = (rows, seen, used, error, closed)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][1] == [0, 2, 4, 6, 8, 10] and results[1][1] == [1, 3, 5, 7, 9, 11]     # the static split
    assert results[0][2] > 1 and results[1][2] > 1                                          # several workers took part
    assert results[0][4] and results[1][4]
    # both ranks reach and leave the collective; rank 1 re-raises its own exception AFTER it, rank 0 learns from the
    # flag row that the table holds sentinels and raises too: nobody continues with a poisoned fitness table
    assert results[1][0] is None and "device lost" in results[1][3]
    assert results[0][0] is None and "rank(s) [1]" in results[0][3]


def _dynamic_failure_worker(rank, world, port, n_candidates, queue):
    import time
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def evaluate(i):
            calls.append(i)
            if rank == 1:
                raise RuntimeError("device lost")     # fails instantly, every time: a sick rank
            time.sleep(0.02)
            return _fake_fitness(i)

        dist.barrier()
        try:
            population.evaluate_population_dynamic(evaluate, n_candidates)
            error = None
        except RuntimeError as exc:
            error = "%s: %s" % (type(exc).__name__, exc)
        queue.put((rank, calls, error))
    finally:
        dist.destroy_process_group()


def test_dynamic_queue_sick_rank_stops_drawing_and_every_rank_raises():
    """a rank whose evaluator fails instantly must not drain the shared queue; all ranks raise."""
    world, n_candidates = 2, 24
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dynamic_failure_worker, args=(r, world, port, n_candidates, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, calls, error = queue.get(timeout=120)
        results[rank] = (calls, error)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(results[1][0]) == 1 and "device lost" in results[1][1]            # one draw, then it stopped
    assert len(results[0][0]) == n_candidates - 1                                 # the healthy rank evaluated the rest
    assert results[0][1].startswith("RemoteEvaluationFailure")
