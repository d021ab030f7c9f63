"""Population-level evaluation: candidates shard embarrassingly across the GPUs of one node, one
process per GPU; the only exchange is one small collective per generation carrying the fitness
scalars — the RCCL-over-xGMI counterpart of the reference's mpi4py object collectives
(optimization/program.py:285-310: allgather/gather helpers; :534-536 round-robin re-evaluation
`i % nprocs == rank`; :582 per-generation `allgather(offspring)`).

Payload: [n_local x (index, convergence factor, time ms, iterations)] float64 — a few KiB, latency
bound; no intra-solve communication ever happens (one solve lives on one GPU).
torch.distributed is used as plumbing only: backend "nccl" is RCCL on ROCm, "gloo" serves the CPU tests.
"""
from typing import Callable, List, Sequence, Tuple


def _dist():
    import torch.distributed as dist
    return dist


def world_info():
    if _native_comm is not None:
        return _native_comm.rank, _native_comm.world
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


_native_comm = None


def use_native_collectives(communicator):
    """Route `allgather_fitness` / `gather_fitness` through the C-ABI communicator (abi.Communicator: evomg_allgather_fitness /
    evomg_gather_fitness, RCCL without torch) instead of torch.distributed; None switches back. The rank / world of the
    communicator take the place of the process group's. The dynamic work queue keeps using the rendezvous store for its
    counter and is not affected."""
    global _native_comm
    _native_comm = communicator


def _group_exists():
    """A process group was initialised — possibly with ONE rank: the collectives then run through it (RCCL on a GPU box)
    exactly as they do with more ranks, instead of being short-circuited."""
    dist = _dist()
    return dist.is_available() and dist.is_initialized()


def shard_indices(n_candidates: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment, as the reference's `i % nprocs == rank` (program.py:534-535)."""
    return [i for i in range(n_candidates) if i % world == rank]


def _device(local_rank):
    import torch
    dist = _dist()
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", local_rank)
    return torch.device("cpu")


def gather_fitness(local: Sequence[Tuple[float, float, float]], world: int, local_rank: int = 0,
                   dst: int = 0):
    """Gather per-rank lists of (rho, time_ms, iterations) to `dst` (north_star: "tiny RCCL gather of the
    fitness scalars back to rank 0"). Every rank must pass the same number of entries. Returns the
    concatenated list on `dst` (rank order), the local list elsewhere. No-op for a single process."""
    if _native_comm is not None:
        rows = [(float(k), float(r[0]), float(r[1]), float(r[2])) for k, r in enumerate(local)]
        table = _native_comm.gather(rows, dst)
        return [tuple(r[1:]) for r in table.tolist()] if table is not None else list(local)
    if world <= 1 and not _group_exists():
        return list(local)
    import torch
    dist = _dist()
    t = torch.tensor(list(local), dtype=torch.float64, device=_device(local_rank)).reshape(-1, 3)
    if dist.get_rank() == dst:
        out = [torch.empty_like(t) for _ in range(world)]
        dist.gather(t, out, dst=dst)
        return [tuple(row) for chunk in out for row in chunk.cpu().tolist()]
    dist.gather(t, None, dst=dst)
    return list(local)


class RemoteEvaluationFailure(RuntimeError):
    """Raised on the HEALTHY ranks of a generation in which another rank's evaluator broke: the gathered table holds
    sentinel rows that are not fitness values, so no rank may continue with it."""


_FLAG_OK, _FLAG_FAILED = -1.0, -2.0


def allgather_fitness(indexed: Sequence[Tuple[int, float, float, float]], n_total: int, world: int,
                      local_rank: int = 0, failed: bool = False):
    """All ranks obtain the fitness of all candidates: rows (index, rho, time_ms, iterations), padded to
    the largest shard. Counterpart of `allgather([(i, fitness)])` (program.py:536). The same collective carries one
    flag row per rank (`failed`: this rank's evaluator broke); if any rank set it, every OTHER rank raises
    RemoteEvaluationFailure after the collective (the failing rank re-raises its own exception)."""
    if _native_comm is not None:
        world = _native_comm.world
        per = (n_total + world - 1) // world
        rows = [[_FLAG_OK] * 4 for _ in range(per + 1)]
        for k, r in enumerate(indexed):
            rows[k] = [float(v) for v in r]
        if failed:
            rows[per][0] = _FLAG_FAILED
        table = _native_comm.allgather(rows).reshape(world, per + 1, 4)
        chunks = [chunk.tolist() for chunk in table]
    elif world <= 1 and not _group_exists():
        return sorted(indexed)
    else:
        import torch
        dist = _dist()
        world = max(world, 1)
        per = (n_total + world - 1) // world
        t = torch.full((per + 1, 4), _FLAG_OK, dtype=torch.float64)
        if indexed:
            t[:len(indexed)] = torch.tensor(list(indexed), dtype=torch.float64)
        if failed:
            t[per, 0] = _FLAG_FAILED
        t = t.to(_device(local_rank))
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        chunks = [chunk.cpu().tolist() for chunk in out]
    sick = [r for r, chunk in enumerate(chunks) if chunk[per][0] == _FLAG_FAILED]
    if sick and not failed:
        raise RemoteEvaluationFailure("evaluator failed on rank(s) %s: this generation's fitness table is invalid" % sick)
    rows = [tuple(r) for chunk in chunks for r in chunk[:per] if r[0] >= 0]
    return sorted((int(r[0]), r[1], r[2], r[3]) for r in rows)


def evaluate_population(evaluate: Callable[[int], Tuple[float, float, float]], n_candidates: int,
                        local_rank: int = 0):
    """Evaluates candidates `i % world == rank` with `evaluate(i) -> (time_ms, rho, iterations)` (the triple
    ProgramGenerator.generate_and_evaluate returns) and returns, on every rank, the list
    [(i, rho, time_ms, iterations)] for ALL candidates."""
    rank, world = world_info()
    mine = []
    failure = None
    for i in shard_indices(n_candidates, rank, world):
        t_ms, rho, its, failure = _guarded(evaluate, i, failure)
        mine.append((i, float(rho), float(t_ms), float(its)))
    try:
        rows = allgather_fitness(mine, n_candidates, world, local_rank, failed=failure is not None)
    finally:
        if failure is not None:
            raise failure
    return rows


FAILED = 1e100      # the reference's `infinity` sentinel (optimization/program.py:71)


def _guarded(evaluate, i, failure):
    """A rank whose evaluator breaks (device error, ...) must still reach the generation's collective, or every other
    rank blocks in it forever: the candidate is recorded with the sentinel triple, the first exception is kept and
    re-raised on this rank AFTER the collective. After a failure the rank evaluates nothing more (its remaining
    candidates get the sentinel without a call: a broken device fails every call at once)."""
    if failure is not None:
        return FAILED, FAILED, FAILED, failure
    try:
        t_ms, rho, its = evaluate(i)
        return t_ms, rho, its, failure
    except Exception as exc:        # noqa: BLE001 — re-raised by the caller once the collective is done
        return FAILED, FAILED, FAILED, failure if failure is not None else exc


_queue_generation = [0]


def _work_queue_store():
    """The process group's key-value store (the TCPStore torch.distributed rendezvoused on): `add` is an atomic
    fetch-and-add served by rank 0's store daemon, which is all a work queue needs."""
    from torch.distributed import distributed_c10d as c10d
    return c10d._get_default_store()


def evaluate_population_dynamic(evaluate: Callable[[int], Tuple[float, float, float]], n_candidates: int,
                                local_rank: int = 0, store=None):
    """Same result as `evaluate_population`, but candidates are handed out from a shared counter instead of the fixed
    `i % world == rank` split: a rank that drew cheap candidates fetches more. Candidates differ ~25x in cost
    (notebooks/tutorial.ipynb:3360: 2-52 ms per iteration), so the static split leaves GPUs idle at the
    end of a generation. The data path still has no collective; per candidate there is one fetch-and-add on the
    rendezvous store (host side, tens of microseconds against solves of tens of milliseconds), and per generation
    one all-reduce of the [n_candidates x 4] fitness table (each row is written by exactly one rank, the others
    contribute zeros, so the sum is exact).
    A rank whose evaluator breaks stops drawing at once (a sick rank fails instantly and would otherwise drain the
    queue, turning candidates the healthy ranks could have evaluated into sentinels); the candidate it held gets the
    sentinel row, the extra last row of the table counts failed ranks, and EVERY rank raises after the collective —
    the failing one its own exception, the others RemoteEvaluationFailure.
    Returns (rows, evaluated_here): rows = [(i, rho, time_ms, iterations)] for all candidates on every rank."""
    rank, world = world_info()
    if world <= 1:
        rows = []
        for i in range(n_candidates):
            t_ms, rho, its = evaluate(i)
            rows.append((i, float(rho), float(t_ms), float(its)))
        return rows, list(range(n_candidates))
    import torch
    dist = _dist()
    store = store if store is not None else _work_queue_store()
    key = "evomg/work_queue/%d" % _queue_generation[0]     # same sequence of calls on every rank -> same key
    _queue_generation[0] += 1
    table = torch.zeros((n_candidates + 1, 4), dtype=torch.float64)
    mine = []
    failure = None
    while failure is None:
        i = store.add(key, 1) - 1
        if i >= n_candidates:
            break
        t_ms, rho, its, failure = _guarded(evaluate, i, failure)
        table[i] = torch.tensor([1.0, float(rho), float(t_ms), float(its)], dtype=torch.float64)
        mine.append(i)
    if failure is not None:
        table[n_candidates, 0] = 1.0
    table = table.to(_device(local_rank))
    dist.all_reduce(table, op=dist.ReduceOp.SUM)
    rows = table.cpu().tolist()
    n_failed = rows.pop()[0]
    if failure is not None:
        raise failure
    if n_failed:
        raise RemoteEvaluationFailure("evaluator failed on %d other rank(s): this generation's fitness table is invalid" % int(n_failed))
    missing = [i for i, r in enumerate(rows) if r[0] != 1.0]
    if missing:
        raise RuntimeError("work queue: candidates %s were evaluated %s" % (missing[:8], "not exactly once"))
    return [(i, r[1], r[2], r[3]) for i, r in enumerate(rows)], mine


def project_ranks(seconds_per_candidate: Sequence[float], ranks: int) -> dict:
    """PROJECTION, not a measurement: wall time of one generation on `ranks` GPUs from per-candidate wall times measured on
    ONE (candidates are independent, one handle per GPU, no data-path collective: SURVEY 8e). static = the reference's split
    i % ranks == rank (optimization/program.py:534-536), dynamic = the shared-counter work queue of
    evaluate_population_dynamic (the next free rank takes the next index). The one collective per generation (:582) is a few
    KiB and is left out; so is any interference between the ranks' host threads (preflight.host_budget).
    tests/test_population_gloo8.py replays a recorded list with eight sleeping `gloo` ranks against these figures."""
    total = float(sum(seconds_per_candidate))
    static = max(sum(seconds_per_candidate[r::ranks]) for r in range(ranks))
    free = [0.0] * ranks
    for t in seconds_per_candidate:
        r = min(range(ranks), key=free.__getitem__)
        free[r] += t
    dynamic = max(free)
    return {"ranks": ranks, "what": "PROJECTION from the per-candidate wall times of the sequential passes on one GPU (no multi-GPU "
                                    "node has been available to any round)",
            "one_gpu_wall_s": total, "static_wall_s": static, "dynamic_wall_s": dynamic,
            "static_speedup": total / static, "dynamic_speedup": total / dynamic,
            "static_efficiency": total / static / ranks, "dynamic_efficiency": total / dynamic / ranks}


MAX_WORKERS_PER_GPU = 16


def warn_if_too_many_workers(n_workers: int) -> None:
    """Sixteen handles per process is the measured optimum AND a limit (profiles/r05_throughput.md): every handle owns a HIP
    stream, `import evostencils_amd` asks the runtime for 24 hardware queues, and with more live streams than those carry every
    launch of the process slows down — the isolated ones as well (18 workers: sequential solves -10 %, the isolated time objective
    off by up to 19 %)."""
    if n_workers > MAX_WORKERS_PER_GPU:
        import warnings
        warnings.warn(f"{n_workers} candidates in flight per GPU: more than {MAX_WORKERS_PER_GPU} live streams slow every launch "
                      f"of the process, isolated solves and time-objective probes included (profiles/r05_throughput.md)", RuntimeWarning, stacklevel=3)


class ConcurrentEvaluator:
    """Throughput mode for ONE GPU: `n_workers` independent ProgramGenerator instances (own HIP stream, own set of
    level fields) driven by as many host threads, so that several candidates are in flight on the device at once.
    Why it pays: a candidate's coarse-grid Krylov solve is one workgroup on one of 256 CUs; while it runs, the other
    workers' bandwidth-bound fine-level sweeps use the rest of the chip. ctypes releases the GIL during evomg_evaluate.

    Convergence factor and iteration count are unaffected (deterministic); the measured time-to-solution of a candidate
    includes contention from its neighbours, so use n_workers=1 (the default everywhere else, and what the reference's
    sequential `map` does, optimization/program.py:491,577) when the time objective must be isolated."""

    def __init__(self, make_generator: Callable[[], object], n_workers: int):
        from concurrent.futures import ThreadPoolExecutor
        import threading
        warn_if_too_many_workers(n_workers)
        self.generators = [make_generator() for _ in range(n_workers)]
        self._pool = ThreadPoolExecutor(max_workers=n_workers)
        self._local = threading.local()
        self._free = list(self.generators)
        self._lock = threading.Lock()

    def _generator(self):
        g = getattr(self._local, "gen", None)
        if g is None:
            with self._lock:
                g = self._free.pop()
            self._local.gen = g
        return g

    def map(self, fn: Callable[[object, object], Tuple[float, float, float]], items: Sequence[object]):
        """fn(generator, item) -> (time_ms, rho, iterations); results in input order."""
        return list(self._pool.map(lambda it: fn(self._generator(), it), items))

    def close(self):
        self._pool.shutdown(wait=True)
        for g in self.generators:
            g.close()


def evaluate_population_in_flight(evaluator: ConcurrentEvaluator, fn: Callable[[object, int], Tuple[float, float, float]],
                                  n_candidates: int, local_rank: int = 0,
                                  time_fn: Callable[[object, int, float, object], float] = None):
    """`evaluate_population` with this rank's share evaluated by the workers of a ConcurrentEvaluator — several
    candidates in flight on the rank's GPU (own generator, stream and fields each) — instead of one after the other.
    fn(generator, i) -> (time_ms, rho, iterations). Same split (`i % world == rank`), same single collective, same
    result table on every rank; a worker that breaks is recorded with the sentinel triple and its first exception is
    re-raised after the collective, as in the sequential form. `time_fn`: see below — with it the table's times are the
    candidates' times ALONE on the GPU (the reference's time objective), without it their contended wall times."""
    rank, world = world_info()
    mine_idx = shard_indices(n_candidates, rank, world)

    def guarded(generator, i):
        try:
            t_ms, rho, its = fn(generator, i)
            # the coarse-solver iteration count of THIS evaluation travels with its result (input of the time model)
            return t_ms, rho, its, getattr(generator, "last_coarse_iterations", None)
        except Exception as exc:        # noqa: BLE001 — re-raised once the collective is done
            return exc

    failure = None
    mine = []
    for i, r in zip(mine_idx, evaluator.map(guarded, mine_idx)):
        if isinstance(r, Exception):
            failure = failure if failure is not None else r
            r = (FAILED, FAILED, FAILED, None)
        t_ms, rho, its, coarse_its = r
        if time_fn is not None and failure is None and t_ms < FAILED and 1 <= its < FAILED:
            # Contention-free time objective: the workers are idle now, so a short probe of the candidate on ONE generator
            # measures it alone — time_fn(generator, i, iterations, coarse_iterations) -> ms (ProgramGenerator.isolated_time_ms
            # with the iteration counts of the in-flight evaluation). The wall time
            # of the in-flight solve, which includes the neighbours, is replaced.
            try:
                t_ms = time_fn(evaluator.generators[0], i, its, coarse_its)
            except Exception as exc:        # noqa: BLE001 — re-raised once the collective is done
                failure = exc
        mine.append((i, float(rho), float(t_ms), float(its)))
    try:
        rows = allgather_fitness(mine, n_candidates, world, local_rank, failed=failure is not None)
    finally:
        if failure is not None:
            raise failure
    return rows


def collective_selftest(local_rank: int = 0):
    """The three collectives of this module (gather, all_gather, all_reduce of small float64 tables) issued directly on
    the initialised process group's device, whatever the world size — with one rank on the `nccl` backend this is what
    runs RCCL's kernels on the GPU without a second device (tests/test_population_rccl_gpu.py). Returns the values
    received, for the caller to check."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = _device(local_rank)
    t = torch.tensor([[float(rank), 0.5, 12.0, 7.0]], dtype=torch.float64, device=dev)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    out = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, out, dst=0)
    table = torch.zeros((4, 4), dtype=torch.float64)
    table[rank % 4] = torch.tensor([1.0, 0.25, 3.0, 9.0], dtype=torch.float64)
    table = table.to(dev)
    dist.all_reduce(table, op=dist.ReduceOp.SUM)
    return {"backend": dist.get_backend(), "device": str(dev), "all_gather": [g.cpu().tolist() for g in gathered],
            "gather": [g.cpu().tolist() for g in out] if out is not None else None, "all_reduce": table.cpu().tolist()}
