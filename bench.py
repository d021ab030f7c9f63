#!/usr/bin/env python3
"""bench.py — candidate multigrid solves per second on the BASELINE.json headline config.

Workload (config.workload, BASELINE.json configs[1]): 2D Poisson 5-point, 4096^2 cells (level 12, 4097^2
nodes), 6-level V(1,1) cycle (levels 12..7) with red-black Gauss-Seidel w=1, CG on the coarsest level,
solved to 1e-12 residual reduction — the reference's problem example_problems/Poisson/2D_FD_Poisson_fromL2.*
at maxLevel=12 evaluated through the drop-in ProgramGenerator.generate_and_evaluate path.

One "step" = one complete candidate evaluation (lower DAG -> op program -> solve on the GPU) with all fields
resident in HBM. With --gpus N each rank evaluates its own candidates (weak scaling, no data-path collective);
the two fitness scalars (convergence factor, time) of every rank are gathered to rank 0 with one tiny RCCL
gather per step, as the optimizer's per-generation allgather does (optimization/program.py:582).

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (a torch.distributed.run
child process, decided before anything touches the GPU) and exits with the child's status.

Besides the headline, the JSON line carries `population`: BASELINE.json configs[4] (256 seeded candidates, 2D Poisson
2048^2, levels 11..7) evaluated once with the static round-robin split (32 per GPU at N = 8) and once through the dynamic
work queue, with per-rank busy/idle time — the numbers the >= 6x @ 8 GPUs target is about (candidates differ ~25x in
cost, so the static split's wall time is its slowest rank).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# hardware queues for the candidates-in-flight modes: must be in the environment before the HIP runtime initialises
# (evostencils_amd/__init__.py explains; ranks started by the launcher inherit it)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
# A/B of the coarse-grid CG's two forms (ProgramGenerator(coarse_cg=...)); the default is the product's default
COARSE_CG = os.environ.get("EVOMG_BENCH_COARSE_CG", "single-reduction")
REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
# HBM-side bytes per launch of the dominant kernel, from rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE separately;
# gfx950 correction FETCH_SIZE x 2): one entry per (kernel family, level) with the profile file it came from. Emitted
# only when the entry matches what this run launched, else null.
TRAFFIC_TABLE = os.path.join(REPO, "profiles", "pmc_traffic.json")


_T0 = time.time()


def progress(stage):
    """One line per stage on stderr (the JSON line owns stdout): where a long run is, and that it is alive."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - _T0:7.1f} s] {stage}", file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--max-level", type=int, default=12)
    ap.add_argument("--min-level", type=int, default=7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-size", action="store_true",
                    help="skip the second configuration (513^2, five levels: the reference's own operating size)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip BASELINE.json configs[2] / configs[3] (Helmholtz 2048^2, 3D Poisson 512^3) on the same line")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--concurrency", type=int, default=1,
                    help="candidates in flight per GPU inside the HEADLINE timed region (1 = isolated timing, the "
                         "reference's sequential semantics)")
    ap.add_argument("--throughput-workers", type=int, default=16,
                    help="after the headline region, also time a throughput-mode region with this many candidates in "
                         "flight per GPU and report it as `throughput_mode` (0/1 = skip)")
    ap.add_argument("--population", type=int, default=256,
                    help="size of the seeded C5 population evaluated after the headline region (0 = skip)")
    ap.add_argument("--population-level", type=int, default=11)
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N outside a torchrun environment: start the N ranks as a child torch.distributed.run (never exec: nothing
    here has touched the GPU, but the child must stay a child) and hand its exit status on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def build_candidate(gen, lo, hi):
    """6-level V(1,1) RB-GS, w = linspace(0.1,1.9,37)[18] (= 1.0 up to round-off, the grammar's value)."""
    from evostencils_amd import candidates
    ctx = candidates.production_context(gen, lo, hi)
    depth = hi - lo
    rb = [("rbgs", 18)]
    expression, _ = candidates.v_cycle(ctx, depth, [rb] * depth, [rb] * depth, cgc_index=18)
    return expression


def cpu_baseline(ops, lo, hi, threads):
    """The oracle (CPU restatement, kind 'port') on ONE full solve of the same workload."""
    os.environ["OMP_NUM_THREADS"] = str(threads)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    from evostencils_amd import problems
    from oracle import oracle
    import ctypes
    # built for THIS host (-march=native); the portable build that travels with the repository otherwise
    try:
        lib, flags = oracle.native_library(), "gcc -O3 -march=native -fopenmp -ffp-contract=off (built on this host)"
    except Exception as exc:        # noqa: BLE001 — no compiler on the box: the baseline is still worth reporting
        lib, flags = None, f"gcc -O3 -fopenmp -ffp-contract=off, portable build (native build failed: {type(exc).__name__})"
    h = oracle.create(problems.poisson_2d(lo, hi), lib=lib)
    t0 = time.time()
    out = h.evaluate(ops, 1, 0, 128)
    wall = time.time() - t0
    # the reference's own setting is omp_numThreads = 4 (example_problems/lib/parallelization_pureOmp.knowledge:3)
    wall4 = None
    if threads > 4:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(4)
        t0 = time.time()
        h.evaluate(ops, 1, 0, 128)
        wall4 = time.time() - t0
    h.close()
    # the TUNED CPU solver beside it (oracle/evomg_cpu_tuned.c: the same algorithm written as a CPU code generator writes it —
    # OpenMP rows, natural summation order, compiler-contracted arithmetic; not bit-comparable with the HIP backend): all the
    # cores this process may run on, and the reference's own omp_numThreads = 4
    tuned = None
    try:
        pd = problems.poisson_2d(lo, hi)
        u0, rhs = pd.initial_solution(hi), pd.right_hand_side(hi)
        runs = {}
        usable = cpu_share(threads)
        for nt in (usable, 4):                       # this process's share of the host first; the reference's 4 threads too if that leaves the sample bounded
            if str(nt) in runs or (runs and sum(v["seconds"] for v in runs.values()) > 12.0):
                continue
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(nt))
            _u, hist, k, seconds = oracle.tuned_poisson2d(lo, hi, u0, rhs, maxit=pd.max_iterations)
            runs[str(nt)] = {"solves_per_s": 1.0 / seconds, "seconds": seconds}
        best = max(runs, key=lambda q: runs[q]["solves_per_s"])
        tuned = {"value": runs[best]["solves_per_s"], "unit": "solves/s", "cores": int(best), "kind": "port",
                 "what": "tuned OpenMP C solver, the same algorithm as a CPU code generator writes it (oracle/evomg_cpu_tuned.c); not bit-comparable with the HIP backend",
                 "sample": f"1 full solve of the same workload ({len(hist) - 1} V-cycles, {k} coarse CG iterations)",
                 "build": "oracle/Makefile: gcc -O3 -march=native -fopenmp (contraction on), built on this host",
                 "iterations": len(hist) - 1, "final_over_initial_residual": float(hist[-1] / hist[0]),
                 "by_threads": {q: v["solves_per_s"] for q, v in runs.items()}}
    except Exception as exc:        # noqa: BLE001
        tuned = {"error": repr(exc)}
    # report the faster of the two as the baseline (memory-bound sweeps do not always scale to all host cores)
    best_wall, best_threads = (wall4, 4) if (wall4 is not None and wall4 < wall) else (wall, threads)
    parity = {"value": 1.0 / best_wall, "unit": "solves/s", "cores": best_threads, "kind": "port",
              "sample": f"1 full solve of the same workload ({int(out.n_iterations)} V-cycles, {best_wall:.1f} s wall)",
              "build": f"oracle/Makefile: {flags}; the Krylov half of the oracle sums in the device's reduction-tree order "
                       "(parity infrastructure, not a tuned CPU solver)",
              "iterations": out.n_iterations, "final_res_norm": out.final_res_norm,
              "by_threads": {str(threads): 1.0 / wall, **({"4": 1.0 / wall4} if wall4 else {})}}
    if tuned and tuned.get("value"):
        # the CPU figure a reader should compare with: the tuned OpenMP solver; the bit-comparable parity
        # oracle — slower by construction — one level down
        return {**tuned, **host_cpu(), "parity_oracle": parity}
    return {**parity, **host_cpu(), "tuned": tuned}


def cpu_share(fallback):
    """CPUs this process may actually use (evostencils_amd.preflight.usable_cores: affinity mask and cgroup quota — a GPU box
    shows all 256 hardware threads to a container that is throttled to 16 of them: an OpenMP team of 256 then spends its time
    being descheduled), cut to 32 (the sweeps are memory-bound well before that)."""
    from evostencils_amd import preflight
    try:
        return max(1, min(preflight.usable_cores(), 32))
    except Exception:                                        # noqa: BLE001
        return max(1, min(fallback, 32))


def host_cpu():
    """CPU model, logical CPUs of the host and the share this process may run on (BASELINE.md asks for them beside every
    CPU number)."""
    model = None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count()
    return {"cpu_model": model, "nproc": os.cpu_count(), "nproc_usable": usable, "cpu_share": cpu_share(usable or 1)}


def measured_traffic(kernel_key):
    try:
        table = json.load(open(TRAFFIC_TABLE))
    except (OSError, ValueError):
        return None, None
    entry = table.get(kernel_key)
    return (entry["bytes_per_launch"], entry["source"]) if entry else (None, None)


def population_section(args, dist, torch, rank, world, local_rank, reduce_device):
    """BASELINE.json configs[4]: `--population` seeded candidates (candidates.population, seed 2024) on 2D Poisson
    level 11, five levels, through generate_and_evaluate with default settings (evaluation_samples = 1); once with the
    static `i % world == rank` split (optimization/program.py:534-535), once through the dynamic work queue."""
    from evostencils_amd import candidates, native, population, problems
    from evostencils_amd.code_generation.hip import ProgramGenerator
    context_for = candidates.production_context
    hi = args.population_level
    lo = hi - 4
    gen = ProgramGenerator(problem=problems.poisson_2d(lo, hi), device_id=local_rank, mpi_rank=rank, library=native.library(), coarse_cg=COARSE_CG)
    gen.initialize_code_generation(lo, hi)
    pop = candidates.population(lambda: context_for(gen, lo, hi), hi - lo, 2, args.population, seed=2024)
    busy = [0.0]
    walls = {}                                 # candidate -> wall seconds of its generate_and_evaluate call (last pass)

    def evaluate(i):
        t0 = time.perf_counter()
        out = gen.generate_and_evaluate(pop[i], None, lo, hi, "", evaluation_samples=1)
        walls[i] = time.perf_counter() - t0
        busy[0] += walls[i]
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_ranks(value):
        if world <= 1:
            return [value]
        t = torch.tensor([value], dtype=torch.float64, device=reduce_device)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    evaluate(rank % args.population)           # warm-up: first launch of every kernel of this level window
    in_flight = args.throughput_workers if args.throughput_workers > 1 else 0
    workers = None
    if in_flight:
        def make():
            g2 = ProgramGenerator(problem=problems.poisson_2d(lo, hi), device_id=local_rank, mpi_rank=rank, library=native.library(), coarse_cg=COARSE_CG)
            g2.initialize_code_generation(lo, hi)
            return g2
        workers = population.ConcurrentEvaluator(make, in_flight)
        workers.map(lambda g2, i: g2.generate_and_evaluate(pop[i], None, lo, hi, "", evaluation_samples=1),
                    [(rank + j) % args.population for j in range(in_flight)])          # warm-up of every worker
    modes = {}
    rows = None
    rows_first = None
    for mode in (("static", "dynamic") if world > 1 else ("static", "static_repeat")) + (("static_in_flight", "static_in_flight_isolated_time") if in_flight else ()):
        progress(f"population: {mode}")
        busy[0] = 0.0
        fence()
        t0 = time.perf_counter()
        if mode in ("static", "static_repeat"):
            if mode == "static_repeat":
                # the sequential pass once more (N = 1): what the time objective of one candidate moves by from one
                # sequential evaluation to the next is the floor under every comparison below
                rows_first, walls_first = rows, dict(walls)
            rows = population.evaluate_population(evaluate, args.population, local_rank)
            mine = len(population.shard_indices(args.population, rank, world))
        elif mode in ("static_in_flight", "static_in_flight_isolated_time"):
            tb = time.perf_counter()
            # ..._isolated_time: the same pass followed by the time model's probes, so that the table's
            # times are each candidate's time ALONE on the GPU — what the reference's sequential evaluation measures
            time_fn = None
            if mode == "static_in_flight_isolated_time":
                time_fn = lambda g2, i, its, k: g2.isolated_time_ms(g2.generate_op_program(pop[i], lo, hi), lo, hi, its, coarse_iterations=k)
            rows_f = population.evaluate_population_in_flight(
                workers, lambda g2, i: g2.generate_and_evaluate(pop[i], None, lo, hi, "", evaluation_samples=1),
                args.population, local_rank, time_fn=time_fn)
            busy[0] = time.perf_counter() - tb
            mine = len(population.shard_indices(args.population, rank, world))
            # rho and iteration counts do not depend on what else runs on the GPU (reported, not asserted: a benchmark
            # line with a `false` in it is worth more than no line)
            same = lambda x, y: x == y or (x != x and y != y)
            identical = all(same(a[1], b[1]) and same(a[3], b[3]) for a, b in zip(rows, rows_f))
        else:
            rows, mine_idx = population.evaluate_population_dynamic(evaluate, args.population, local_rank)
            mine = len(mine_idx)
        fence()
        wall = max(all_ranks(time.perf_counter() - t0))
        busy_all, count_all = all_ranks(busy[0]), all_ranks(float(mine))
        modes[mode] = {"solves_per_s": args.population / wall, "wall_s": wall,
                       "busy_s_per_rank": [round(b, 4) for b in busy_all],
                       "idle_s_per_rank": [round(wall - b, 4) for b in busy_all],
                       "candidates_per_rank": [int(c) for c in count_all]}
        if mode in ("static_in_flight", "static_in_flight_isolated_time"):
            modes[mode]["candidates_in_flight_per_gpu"] = in_flight
            modes[mode]["rho_and_iterations_identical_to_sequential"] = bool(identical)
            # the time objective of this mode against the sequential pass's (finite candidates of THIS rank's share are all
            # candidates at N = 1; at N > 1 every rank holds the whole table)
            # reference time of a candidate: the faster of its two sequential solves where there are two (a disturbed solve
            # is slower, never faster)
            ref = [min(a[2], c[2]) for a, c in zip(rows, rows_first)] if rows_first is not None else [a[2] for a in rows]
            pairs = [(t, b[2]) for t, b in zip(ref, rows_f) if t < 1e99 and b[2] < 1e99 and t > 0]
            if pairs:
                modes[mode]["time_objective_vs_sequential"] = _compare_times([p[0] for p in pairs], [p[1] for p in pairs])
        if mode == "static_repeat":
            pairs = [(a[2], c[2]) for a, c in zip(rows_first, rows) if a[2] < 1e99 and c[2] < 1e99 and a[2] > 0]
            modes[mode]["sequential_pass_vs_itself"] = _compare_times([p[0] for p in pairs], [p[1] for p in pairs])
            modes[mode]["note"] = ("the sequential pass a second time: how far one candidate's time moves between two sequential "
                                   "evaluations on this box — the floor under time_objective_vs_sequential (whose reference is the "
                                   "faster of the two)")
            per_candidate = [min(walls_first[i], walls[i]) for i in range(args.population)]
            summary["projected_8gpu"] = _project_ranks(per_candidate, 8)
            # the times the projection is made from (ms per generate_and_evaluate call, candidate order): the fixture of the
            # eight-rank rehearsal (tests/golden/c5_candidate_times.json) is one such list
            summary["per_candidate_ms"] = [round(1e3 * t, 3) for t in per_candidate]
        if mode == "static":
            finite = [r for r in rows if r[3] < 1e99]
            costs = sorted(r[2] for r in rows)
            summary = {"converged_below_cap": sum(1 for r in rows if r[3] < gen.problem.max_iterations and r[1] < 1),
                       "finite": len(finite), "best_rho": min(r[1] for r in rows),
                       "solve_ms_min_median_max": [costs[0], costs[len(costs) // 2], costs[-1]]}
    gen.close()
    if workers is not None:
        workers.close()
    return {"workload": f"G3P population = {args.population} seeded candidates (numpy default_rng(2024)), 2D Poisson "
                        f"{1 << hi}^2 (level {hi}), levels {hi}..{lo}, evaluation_samples = 1, "
                        f"jacobi_semantics = {gen.jacobi_semantics}",
            "n_gpus": world, **summary, **modes,
            "note": "static = round-robin i % N == rank (the reference's split); dynamic = shared counter on the rendezvous "
                    "store, one all-reduce of the fitness table per generation; idle = wall - time spent inside "
                    "generate_and_evaluate on that rank. At N = 1 both modes coincide. static_in_flight = the static split with "
                    "several candidates in flight per GPU (population.ConcurrentEvaluator: own stream and fields each; rho and "
                    "iteration counts identical — checked, see the flag — while a candidate's measured time includes its neighbours). "
                    "static_in_flight_isolated_time = the same, then short probes per candidate on the idle GPU (evomg_time_model for a worker's first three candidates, evomg_time_model_known_coarse after them: what one coarse-solver iteration costs does not depend on the candidate) and "
                    "the evaluation's own iteration counts give each candidate's time ALONE: time_objective_vs_sequential compares it with "
                    "the sequential pass (static)."}


def _compare_times(reference, other):
    """How a list of per-candidate times compares with a reference list: ratios, deviations, rank correlation."""
    import numpy as np
    ts, tm = np.array(reference, float), np.array(other, float)
    order = lambda v: np.argsort(np.argsort(v)).astype(float)
    ra, rb = order(ts) - order(ts).mean(), order(tm) - order(tm).mean()
    rel = np.abs(tm - ts) / ts
    return {"median_ratio": float(np.median(tm / ts)), "max_rel_deviation": float(rel.max()),
            "p99_rel_deviation": float(np.percentile(rel, 99)), "candidates_beyond_5_percent": int(np.sum(rel > 0.05)),
            "spearman": float((ra * rb).sum() / np.sqrt((ra * ra).sum() * (rb * rb).sum())), "candidates": int(len(ts))}


def _project_ranks(seconds_per_candidate, ranks):
    """PROJECTION, not a measurement (evostencils_amd.population.project_ranks; rehearsed with eight `gloo` ranks replaying a
    recorded set of per-candidate times in tests/test_population_gloo8.py)."""
    from evostencils_amd import population
    return population.project_ranks(seconds_per_candidate, ranks)


def reference_size_section(local_rank, rank, steps=20, warmup=3, in_flight=0):
    """Second configuration (rank 0, N = 1): the size the reference's own optimisation runs at — 2D Poisson level 9
    (513^2), five levels, `notebooks/tutorial.ipynb` — with the headline's V(1,1) red-black cycle. Small grids are
    launch- and latency-bound: reported per V-cycle, not against the HBM roofline."""
    from evostencils_amd import native, problems
    from evostencils_amd.code_generation.hip import ProgramGenerator
    lo, hi = 5, 9
    gen = ProgramGenerator(problem=problems.poisson_2d(lo, hi), device_id=local_rank, mpi_rank=rank, library=native.library(), coarse_cg=COARSE_CG)
    gen.initialize_code_generation(lo, hi)
    storages = gen.generate_storage(lo, hi, gen.finest_grid)
    expression = build_candidate(gen, lo, hi)
    for _ in range(warmup):
        gen.generate_and_evaluate(expression, storages, lo, hi, "", evaluation_samples=1)
    t0 = time.perf_counter()
    for _ in range(steps):
        t_ms, rho, its = gen.generate_and_evaluate(expression, storages, lo, hi, "", evaluation_samples=1)
    wall = time.perf_counter() - t0
    gen.close()
    flight = None
    if in_flight > 1:
        # the same candidate evaluated by `in_flight` workers at once (own generator, stream and fields each): what a
        # population run at this size sees. Lowering runs under the interpreter lock, the solves do not.
        from evostencils_amd import population

        def make():
            g = ProgramGenerator(problem=problems.poisson_2d(lo, hi), device_id=local_rank, mpi_rank=rank, library=native.library(), coarse_cg=COARSE_CG)
            g.initialize_code_generation(lo, hi)
            return g
        workers = population.ConcurrentEvaluator(make, in_flight)
        cands = [build_candidate(g, lo, hi) for g in workers.generators]
        by_gen = {id(g): c for g, c in zip(workers.generators, cands)}

        def one(g, _):
            return g.generate_and_evaluate(by_gen[id(g)], None, lo, hi, "", evaluation_samples=1)
        workers.map(one, range(4 * in_flight))
        n = 40 * in_flight
        t1 = time.perf_counter()
        out = workers.map(one, range(n))
        wall2 = time.perf_counter() - t1
        workers.close()
        flight = {"candidates_in_flight_per_gpu": in_flight, "steps": n, "value": n / wall2, "unit": "solves/s",
                  "iterations_identical": all(r[2] == its for r in out)}
    return {"workload": f"2D Poisson 5-pt, {1 << hi}^2 grid (level {hi}), {hi - lo + 1}-level V(1,1) RB-Gauss-Seidel cycle, "
                        "CG coarse solve on 33^2 to 1e-12",
            "in_flight": flight,
            "other_examples": reference_examples(local_rank),
            "steps": steps, "value": steps / wall, "unit": "solves/s", "ms_per_step": 1e3 * wall / steps,
            "solve_time_ms": t_ms, "iterations_per_solve": its, "convergence_factor": rho,
            "ms_per_vcycle": t_ms / its if its else None,
            "reference_recorded": {"ms_per_iteration": 367.47433333333333 / 100.0,
                                   "note": "notebooks/tutorial.ipynb:3373: the reference's best evolved cycle (a different, "
                                           "cheaper-per-iteration candidate) on its CPU, 100 iterations in 367.5 ms at this "
                                           "size; context, not a like-for-like baseline"},
            "note": "solve_time_ms = device time of the solver loop (what generate_and_evaluate returns); ms_per_step = wall "
                    "clock per generate_and_evaluate call incl. lowering and the host side of the launches"}


def reference_examples(local_rank):
    """The reference's other example problems at the level ranges its own files set (example_problems/*/*.knowledge), each
    with the problem's own solver settings as the candidate: one solve alone on the device, best of three."""
    from evostencils_amd import abi, candidates, native, problems
    out = {}
    cases = (("helmholtz_2d_129", "2D Helmholtz 128^2 (levels 3..7, k = 80), BiCGStab preconditioned with the hand-written V(2,1) "
              "(Helmholtz/2D_FD_Helmholtz_fromL3.exa3:202-272)", problems.helmholtz_2d(3, 7, k=80.0),
              candidates.v_cycle_ops(3, 7, 2, 1, omega=0.6)),
             ("poisson_3d_65", "3D Poisson 64^3 (levels 2..6), V(2,1) red-black w = 1.25, CG on 5^3 "
              "(Poisson/3D_FD_Poisson_fromL2.exa3)", problems.poisson_3d(2, 6),
              candidates.v_cycle_ops(2, 6, 2, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK,
                                                                       (1, 1, 1), 1.25))))
    for key, what, pd, ops in cases:
        h = native.create(pd, device_id=local_rank)
        try:
            h.evaluate(ops, 1, 0, 0)
            runs = [h.evaluate(ops, 1, 0, 0) for _ in range(3)]
            best = min(runs, key=lambda r: r.time_ms)
            out[key] = {"workload": what, "solve_time_ms": best.time_ms, "iterations": best.n_iterations, "status": best.status,
                        "ms_per_iteration": best.time_ms / best.n_iterations if best.n_iterations else None}
        finally:
            h.close()
    return out


def _frac(b):
    return {"ms_per_launch": b.ms, "bytes_per_launch": b.bytes, "launches": b.launches, "achieved_gbps": b.gbps,
            "frac": b.gbps / 8000.0}


def other_configs_section(local_rank):
    """BASELINE.json configs[2] and configs[3] on the driver's clock (rank 0, N = 1): the same live HIP-event timing as the
    headline's roofline (evomg_profile_begin/end on the backend's stream, fused launches timed as what they are).
    Algorithmic bytes per launch (complex x2): sweep 24 B/node (+ 8 B per coarse node where the coarse level is
    read or written by the same launch; - 8 B/node for a sweep that starts from an elided zero field), 3D double sweep
    24 B/node for BOTH sweeps, residual + restriction 16 B/node + 8 B per coarse node."""
    import dataclasses
    from evostencils_amd import abi, native, problems
    from evostencils_amd import candidates
    vcycle = candidates.v_cycle_ops
    out = {}
    # ---- C3: 2D Helmholtz 2048^2 (complex), k h = 0.625, BiCGStab preconditioned with the EVOLVED cycle (BASELINE.json
    # configs[2]: candidates.evolved_helmholtz_double_cycle — the cycle tests/test_baseline_configs_gpu.py checks against the
    # oracle at this size for these thirty iterations), capped; the reference's hand-written V(2,1) beside it
    lo, hi, cap = 7, 11, 30
    gen3, ops = candidates.evolved_helmholtz_double_cycle(lo, hi, 0.625 * (1 << hi))
    pd = dataclasses.replace(gen3.problem, max_iterations=cap)
    h = native.create(pd, local_rank)
    h.evaluate(ops, 1, 0, 0)                                  # warm-up: first launch of every kernel
    h.profile_begin(abi.OP_SMOOTH, hi)
    res = h.evaluate(ops, 1, 0, 0)
    sweeps = h.profile_end()
    smooth = next(o for o in ops if o.opcode == abi.OP_SMOOTH and o.level == hi)
    hand = candidates.v_cycle_ops(lo, hi, 2, 1, omega=0.6)
    h.evaluate(hand, 1, 0, 0)
    res_hand = h.evaluate(hand, 1, 0, 0)
    out["c3"] = {"workload": f"2D Helmholtz, {1 << hi}^2 (complex128), k = {pd.parameters['k']:g} (k h = 0.625), levels {hi}..{lo}, "
                             f"MG-preconditioned BiCGStab (2 preconditioner applications per iteration), EVOLVED cycle = two V(2,1) "
                             f"RB-GS w = 0.6 corrections per application (4 V-cycles per outer iteration), capped at {cap} outer "
                             f"iterations (no cycle converges inside the reference's cap at this size, DESIGN section 5)",
                 "ms_per_outer_iteration": res.time_ms / res.n_iterations, "outer_iterations": res.n_iterations,
                 "solve_time_ms": res.time_ms,
                 "reference_vcycle": {"what": "the reference's hand-written V(2,1) cycle as preconditioner (exa3:202-272: 2 V-cycles "
                                              "per outer iteration), same cap", "ms_per_outer_iteration": res_hand.time_ms / res_hand.n_iterations},
                 "dominant_kernel": {"name": h.kernel_name(smooth), "what": "finest-level RB-GS sweeps as this timed pass launches them (12 per outer iteration: "
                                     "plain, sweep + residual + restriction, prolongation + sweep; the pass times one launch per SMOOTH op, "
                                     "so the round-6 fusion of a pre-smoothing pair that starts from zero into ONE launch - which the "
                                     "untimed solves above run - is off here)", **_frac(sweeps),
                                     "share_of_solve": sweeps.ms * sweeps.launches / res.time_ms}}
    h.close()
    # ---- C4: 3D Poisson 512^3, V(2,2) block-Jacobi (2,1,1) w = 0.8, levels 9..5, textbook `with jacobi` semantics
    lo, hi = 5, 9
    pd = problems.poisson_3d(lo, hi)
    h = native.create(pd, local_rank, jacobi_semantics="textbook")
    ops = vcycle(lo, hi, 2, 2, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2, 1, 1), 0.8))
    h.evaluate(ops, 1, 0, 0)
    kernels = {}
    solve = None
    for key, opcode, what in (("double_sweep", abi.OP_SMOOTH, "two block-Jacobi (2,1,1) sweeps in ONE pass (evomg_tile3d_twice; the "
                                                               "post-smoothing launch also folds the prolongation in)"),
                              ("residual_restrict", abi.OP_RESIDUAL, "residual + full weighting, residual never written "
                                                                     "(evomg_tile3d_residual_restrict)"),
                              ("residual_norm", abi.PROF_LOOP_NORM, "the solver loop's residual norm: it rides on the COMING cycle's first "
                                                                     "double sweep, launched speculatively (evomg_tile3d_twice<NORM0>: x, f read "
                                                                     "once for the norm and both sweeps, the swept field written) + two "
                                                                     "reduction launches; this launch is also one of double_sweep's"),
                              ("coarse_solve", abi.OP_COARSE_SOLVE, "CG on 33^3, single-reduction form on eight rendezvousing workgroups of one XCD (evomg_cg1_slab3d)")):
        h.profile_begin(opcode, lo if opcode == abi.OP_COARSE_SOLVE else hi)
        solve = h.evaluate(ops, 1, 0, 0)
        b = h.profile_end()
        kernels[key] = {"what": what, **_frac(b), "share_of_solve": b.ms * b.launches / solve.time_ms}
        if opcode == abi.OP_COARSE_SOLVE:
            kernels[key].pop("frac"), kernels[key].pop("achieved_gbps"), kernels[key].pop("bytes_per_launch")
    rb = abi.make_op(abi.OP_SMOOTH, hi, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.0)
    b2b = sorted((h.bench_op(rb, warmup=3, reps=20) for _ in range(3)), key=lambda b: b.ms)[1]
    kernels["rbgs_back_to_back"] = {"what": "fused two-colour RB-GS sweep (evomg_tile3d_rbgs), 20 back-to-back launches, median of 3",
                                    **_frac(b2b)}
    out["c4"] = {"workload": f"3D Poisson 7-pt, {1 << hi}^3 (level {hi}), levels {hi}..{lo}, V(2,2) block-Jacobi (2,1,1) w = 0.8, "
                             "CG on 33^3, 1e-12 reduction; jacobi_semantics = textbook (under the reference pipeline's semantics the "
                             "sweeps never take effect and the candidate does not converge: tests/test_baseline_configs_gpu.py)",
                 "ms_per_solve": solve.time_ms, "iterations": solve.n_iterations, "convergence_factor": solve.conv_factor,
                 "status": solve.status, "kernels": kernels}
    h.close()
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    # ONE JSON line on stdout, nothing else: RCCL prints a version banner to stdout when its first communicator comes up
    # (one-rank group at N = 1 included), so everything below writes to stderr's file and only the line goes to the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report n_gpus != --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    # EVOMG_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives on host tensors); the real multi-GPU run uses nccl (= RCCL), one rank per GPU
    backend = os.environ.get("EVOMG_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} visible GPUs, found {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    rccl = {"backend": backend, "ranks": world, "error": None}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif backend == "nccl":
        # N = 1: a ONE-rank RCCL group, so that the per-step fitness gather and the population's per-generation collective
        # run through RCCL here exactly as they do at N > 1 (population.py goes through the group whenever one exists)
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as sock:
                    sock.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        except Exception as exc:                              # noqa: BLE001 — must not cost the headline line
            rccl["error"] = repr(exc)
    reduce_device = "cuda" if backend == "nccl" else "cpu"

    from evostencils_amd import abi, native, problems
    from evostencils_amd.code_generation.hip import ProgramGenerator
    from evostencils_amd import population

    # what tools/preflight_8gpu.py checks without its --collective part, on every rank of THIS launch: queue / IPC environment,
    # device count, LOCAL_RANK -> device, librccl (the collective that matters here is the process group above; the C-ABI
    # communicator's own all-gather across ranks stays with the preflight tool: a first contact with a multi-GPU node
    # belongs in a tool that may hang, not in the start-up of the benchmark)
    try:
        from evostencils_amd import preflight
        rccl["preflight"] = preflight.report(collective=False)
    except Exception as exc:                                  # noqa: BLE001 — reported on the line, must not cost it
        rccl["preflight"] = {"ok": False, "failures": [repr(exc)]}

    lo, hi = args.min_level, args.max_level
    gen = ProgramGenerator(problem=problems.poisson_2d(lo, hi), device_id=local_rank, mpi_rank=rank,
                           library=native.library(), coarse_cg=COARSE_CG)
    gen.initialize_code_generation(lo, hi)
    storages = gen.generate_storage(lo, hi, gen.finest_grid)
    expression = build_candidate(gen, lo, hi)

    workers = None
    n_extra = max(args.concurrency, args.throughput_workers if args.throughput_workers > 1 else 1) - 1
    if n_extra > 0:
        def make():
            g2 = ProgramGenerator(problem=problems.poisson_2d(lo, hi), device_id=local_rank, mpi_rank=rank,
                                  library=native.library(), coarse_cg=COARSE_CG)
            g2.initialize_code_generation(lo, hi)
            return g2
        workers = population.ConcurrentEvaluator(make, n_extra)
        expressions = [build_candidate(g2, lo, hi) for g2 in workers.generators]
        import threading
        extra_lock = threading.Lock()

    def evaluate_one():
        t, rho, its = gen.generate_and_evaluate(expression, storages, lo, hi, "", evaluation_samples=1)
        return (rho, t, its)

    def step():
        return population.gather_fitness([evaluate_one()], world, local_rank)

    def run_steps(n, in_flight=None):
        """n candidate evaluations with `in_flight` candidates on the GPU at once (extra workers pull from the same
        budget of n evaluations)."""
        in_flight = args.concurrency if in_flight is None else in_flight
        if workers is None or in_flight <= 1:
            last = None
            for _ in range(n):
                last = step()
            return last
        import concurrent.futures as cf
        remaining = [n]

        def take():
            with extra_lock:
                if remaining[0] <= 0:
                    return False
                remaining[0] -= 1
                return True

        def worker(g2, e2):
            while take():
                g2.generate_and_evaluate(e2, None, lo, hi, "", evaluation_samples=1)

        last = None
        with cf.ThreadPoolExecutor(max_workers=len(workers.generators)) as pool:
            futs = [pool.submit(worker, g2, e2) for g2, e2 in list(zip(workers.generators, expressions))[:in_flight - 1]]
            # no collective inside this loop: how many of the n evaluations the main thread gets depends on thread
            # timing and differs between ranks; the fitness gather happens once, after the pool has drained
            while take():
                last = evaluate_one()
            for f_ in futs:
                f_.result()
        return population.gather_fitness([last if last is not None else evaluate_one()], world, local_rank)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ops = gen.generate_op_program(expression, lo, hi)
    smooth = next(o for o in ops if o.opcode == abi.OP_SMOOTH and o.level == hi)
    handle = gen._handle

    progress("warm-up")
    run_steps(args.warmup)
    progress("headline: timed steps")
    fence()
    # live HIP-event timing of the dominant kernel (finest-level RB-GS sweep) on the backend's own stream,
    # over exactly the timed region
    handle.profile_begin(abi.OP_SMOOTH, hi)
    t0 = time.perf_counter()
    last = run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    live = handle.profile_end()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- roofline of the dominant kernel: finest-level fused RB-GS sweep, 24 B/node algorithmic ----
    kname = handle.kernel_name(smooth)
    n_nodes = ((1 << hi) + 1) ** 2
    plain_bytes = 24.0 * n_nodes
    fused = live.bytes > 1.001 * plain_bytes
    cross = live.bytes > 1.12 * plain_bytes        # the cross-iteration launch: 24 B/node + two coarse fields (x 1/4 each)
    traffic, traffic_source = measured_traffic(f"strip2d_cross_L{hi}" if cross else f"strip2d_rbgs_{'fused' if fused else 'plain'}_L{hi}")
    if cross:
        # since round 4 one launch per iteration does the cycle's last sweep (+ prolongation, + the loop's norm) AND the next
        # cycle's first sweep (+ residual + restriction); the first iteration's head and the last one's tail run as the
        # round-3 fused sweeps (they are in the mean below: 2 of 13 launches per solve)
        kname = ("evomg_strip2d_cross<true, true, true, true, true> (csrc/evomg_strip_cross.h; the last flag: unit-weight form): prolongation + sweep + residual norm of "
                 "iteration k and sweep + residual + restriction of iteration k + 1 in ONE pass; replaces two launches of "
                 f"{(24.0 * n_nodes + 8.0 * ((1 << (hi - 1)) + 1) ** 2) / 1e6:.1f} MB each")
    elif fused:
        # inside a solve both finest-level sweeps of the V(1,1) cycle run fused with their neighbours: the
        # pre-smoother with residual + restriction, the post-smoother with prolongation and the loop's residual norm
        kname = (kname.replace(">", ", ...>") + " fused sweeps: <.., false, false, true> = smoother + residual + restriction, "
                 "<.., true, true, false> = prolongation + smoother + residual norm")
    # SURVEY 8(d) procedure: >= 10 warm-up launches, 100 timed back-to-back ping-pong sweeps, median of 5 batches
    b2b = sorted((handle.bench_op(smooth, warmup=10, reps=100) for _ in range(5)), key=lambda b: b.ms)[2]
    roofline = {"bound": "hbm", "achieved": live.gbps, "peak": 8000.0, "unit": "GB/s", "frac": live.gbps / 8000.0,
                "traffic": traffic, "traffic_source": traffic_source, "kernel": kname,
                "bytes_per_launch": live.bytes, "ms_per_launch": live.ms, "launches_timed": live.launches,
                "back_to_back": {"kernel": handle.kernel_name(smooth), "bytes_per_launch": plain_bytes,
                                 "ms_per_launch": b2b.ms, "achieved": b2b.gbps, "frac": b2b.gbps / 8000.0},
                "finest_level_bytes_per_iteration": {"now": live.bytes * live.launches / (args.steps * float(last[0][2])) if cross else None,
                                                     "round3_two_fused_launches": 2 * (24.0 * n_nodes + 8.0 * ((1 << (hi - 1)) + 1) ** 2),
                                                     "unit": "algorithmic B per solver-loop iteration"},
                "note": f"'achieved' = algorithmic bytes / mean duration over the {live.launches} launches that execute the "
                        f"finest-level smoothing ops inside the timed solves (HIP events on the backend stream). "
                        f"Algorithmic bytes per launch: 24 B/node x {n_nodes} nodes for a plain fused red+black sweep; "
                        f"the fused sweeps additionally write / read the coarse level (8 B per coarse node); the cross-iteration "
                        f"launch reads x, f and the coarse correction and writes w and the coarse right-hand side "
                        f"(24 B/node + 2 x 8 B per coarse node) for the work of TWO sweeps; mean over the timed launches: "
                        f"{live.bytes / 1e6:.1f} MB. 'back_to_back' = median of 5 batches of 100 consecutive plain ping-pong sweeps. "
                        f"'traffic' = HBM-side bytes/launch from rocprofv3 PMC passes (profiles/pmc_traffic.json names the "
                        f"profile), null when no profile matches the launched kernel and level. What holds the cross launch at "
                        f"0.60-0.66 of the HBM peak (round 6, profiles/r06_cross_kernel.md): two walls at the same place. It moves "
                        f"its 534 MB at the fabric at 6.0-6.2 TB/s - the copy rate the hardware guide measures (6.29) - and a wave "
                        f"marches its 91 row steps at 0.95 us each, a chain of dependent stages: the unit-weight form C2 runs since "
                        f"round 6 (19 % fewer FP64 instructions per row step) and prefetch rings of 3 and 6 rows take the same "
                        f"time, taller chunks (fewer halo rows, fewer bytes) take LONGER (96 .. 144 rows: 95 .. 128 us)"}

    # ---- where the rest of an isolated solve goes: the coarse-grid solve, timed the same way over three more solves ----
    progress("coarse solve timing")
    handle.profile_begin(abi.OP_COARSE_SOLVE, lo)
    extra = run_steps(3)
    fence()
    cs = handle.profile_end()
    coarse_op = next(o for o in ops if o.opcode == abi.OP_COARSE_SOLVE)
    coarse_solve = {"kernel": handle.kernel_name(coarse_op), "grid": f"{(1 << lo) + 1}^2", "ms_per_call": cs.ms,
                    "calls_per_solve": cs.launches / 3.0, "share_of_solve": cs.ms * cs.launches / 3.0 / float(extra[0][1]),
                    "iterations_per_call": (handle.coarse_iterations() / (cs.launches / 3.0)) if cs.launches else None,
                    "bound": "one grid-wide exchange per CG iteration (a round trip through one XCD's L2) + three dependent reduction / scalar chains"
                             if COARSE_CG == "single-reduction" else "VALU issue of ONE compute unit",
                    "note": "CG to 1e-12 on the coarsest grid (the reference's solver settings, Poisson/2D_FD_Poisson_fromL2.exa3:12-14). "
                            "The single-reduction form of the recurrence (Chronopoulos-Gear: both inner products of an "
                            "iteration in ONE reduction; csrc/evomg_cg1.h) on sixteen co-resident workgroups of two waves - the five "
                            "vectors of 129^2 in registers, ghost rows instead of a second exchange, exchanged doubles as self-validating "
                            "16-byte words that stay in ONE XCD's L2, every wave exchanging for itself: 0.96 us per iteration (1.47 before "
                            "round 6 read the loop's ISA: exchange addresses formed once, wave-uniform polls, the first look timed behind "
                            "the stores, one division on the scalar chain instead of square root + two: profiles/r06_cg1.md) where the "
                            "classical two-reduction form on ONE compute unit "
                            "(EVOMG_BENCH_COARSE_CG=classical: cgs_kind CGS_CG_CLASSICAL, evomg_cg_reg2d) takes 2.05; of an iteration's "
                            "~2 300 cycles about 600 are the L2 round trip of the exchange, the rest the ~375 instructions a wave issues "
                            "(88 of them the stencil's arithmetic) and four dependent chains: two reduction trees, the scalars, stencil + update"}
    # the whole solve against the roofline of the metric BASELINE names (candidate solves/s): algorithmic bytes of one solve as
    # launched — per iteration the cross launch on the finest level (24 B/node + 2 x 8 B per coarse node), on every level between
    # a pre-smoothing launch from an elided zero field with residual + restriction (16 + 2 B/node) and a prolongation + sweep
    # (24 + 2 B/node); the first iteration's head and the last one's tail as two plain fused launches more
    _rho_h, t_ms, its = (float(v) for v in last[0])
    nodes = lambda L: float(((1 << L) + 1) ** 2)
    bytes_per_iteration = 28.0 * nodes(hi) + sum(44.0 * nodes(L) for L in range(lo + 1, hi))
    bytes_per_solve = bytes_per_iteration * its + 2.0 * 26.0 * nodes(hi)
    solve_roofline = {"bound": "hbm", "algorithmic_bytes_per_solve": bytes_per_solve, "algorithmic_bytes_per_iteration": bytes_per_iteration,
                      "achieved": bytes_per_solve / (t_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                      "frac": bytes_per_solve / (t_ms * 1e-3) / 8e12,
                      "note": "ONE candidate alone on the GPU: its coarse-grid solves (latency-bound, 8 of 256 compute units) are most "
                              "of its time; the bandwidth-bound rate of the metric is throughput_mode.roofline"}

    progress("throughput mode")
    throughput = None
    if args.throughput_workers > 1 and args.throughput_workers != args.concurrency:
        k = args.throughput_workers
        n = max(args.steps, 24 * k)               # long enough for the ramp at both ends not to weigh (16 in flight: 384 solves, a second; with 192 the last, partly idle round was a twelfth of the region)
        run_steps(k, in_flight=k)
        fence()
        t1 = time.perf_counter()
        run_steps(n, in_flight=k)
        fence()
        el2 = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([el2], dtype=torch.float64, device=reduce_device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el2 = float(tt.item())
        rate = n / el2                            # per GPU
        throughput = {"candidates_in_flight_per_gpu": k, "steps": n, "value": world * n / el2, "unit": "solves/s",
                      "ms_per_step": 1e3 * el2 / n,
                      "roofline": {"bound": "hbm", "algorithmic_bytes_per_solve": bytes_per_solve, "achieved": bytes_per_solve * rate / 1e9,
                                   "peak": 8000.0, "unit": "GB/s", "frac": bytes_per_solve * rate / 8e12,
                                   "traffic_per_solve": measured_traffic("c2_solve")[0],
                                   "note": "per GPU: algorithmic bytes of one solve x solves per second. With a dozen working sets of "
                                           "1.1 GB each nothing stays in the 256 MB Infinity Cache: the streaming rate these kernels "
                                           "reach from HBM alone is 0.55-0.57 of peak (8193^2, profiles/r04_8193_raw.txt), which puts the "
                                           "ceiling of this mode near 500 solves/s; profiles/r05_throughput.md has the kernel trace "
                                           "(what is in flight when) and the counter passes"},
                      "note": "same workload, several candidates in flight per GPU (own stream + fields each): the latency-bound "
                              "coarse-grid solve of one candidate overlaps the bandwidth-bound sweeps of the others; rho/iterations "
                              "unchanged, per-candidate time includes contention"}
    if workers is not None:
        workers.close()

    pop_section = None
    if args.population > 0:
        progress("population")
        pop_section = population_section(args, dist, torch, rank, world, local_rank, reduce_device)

    if rank == 0:
        rho, t_ms, its = last[0]
        line = {
            "metric": "candidate MG solves/sec", "value": world * args.steps / elapsed, "unit": "solves/s",
            # the same metric with the GPU full (BASELINE's "whole node" rate): throughput_mode's figures, repeated up here so
            # that a record which keeps only the head of the line carries them (VERDICT r5 #1d)
            "value_in_flight": throughput["value"] if throughput is not None else None,
            "roofline_in_flight": ({"frac": throughput["roofline"]["frac"], "achieved": throughput["roofline"]["achieved"],
                                    "peak": 8000.0, "unit": "GB/s", "candidates_in_flight_per_gpu": throughput["candidates_in_flight_per_gpu"]}
                                   if throughput is not None else None),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"2D Poisson 5-pt, {1 << hi}^2 grid (level {hi}), {hi - lo + 1}-level V(1,1) "
                                   f"RB-Gauss-Seidel cycle, CG coarse solve, 1e-12 reduction",
                       "iterations_per_solve": its, "convergence_factor": rho, "solve_time_ms": t_ms,
                       "parallelism": f"candidates sharded 1/GPU x {world}, "
                                      + ("RCCL fitness gather" if (dist.is_initialized() and backend == "nccl") else
                                         f"fitness gather: {'no process group' if not dist.is_initialized() else backend}"),
                       "collectives": {**rccl, "process_group": dist.is_initialized(),
                                       "n_ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,
                                       "per_step": "gather of (rho, time, iterations) to rank 0 inside the timed region",
                                       "per_generation": "all_gather of [n_local x (index, rho, time, iterations)] + failure flag row "
                                                         "in the population section"},
                       "candidates_in_flight_per_gpu": args.concurrency},
            "roofline": roofline,
            "solve_roofline": solve_roofline,
            "coarse_solve": coarse_solve,
        }
        if throughput is not None:
            line["throughput_mode"] = throughput
        if pop_section is not None:
            line["population"] = pop_section
        if world == 1 and not args.no_reference_size:
            try:                                             # auxiliary, rank-local: must not cost the headline line
                progress("reference size")
                line["reference_size"] = reference_size_section(local_rank, rank, in_flight=args.throughput_workers)
            except Exception as exc:                         # noqa: BLE001
                line["reference_size"] = {"error": repr(exc)}
        if world == 1 and not args.no_configs:
            try:                                             # auxiliary, rank-local: must not cost the headline line
                progress("configs C3, C4")
                line["configs"] = other_configs_section(local_rank)
            except Exception as exc:                         # noqa: BLE001
                line["configs"] = {"error": repr(exc)}
        if not args.no_cpu_baseline and world == 1:         # the CPU baseline is reported at N = 1 only
            threads = args.cpu_threads or min(16, cpu_share(os.cpu_count() or 1))
            try:
                progress("cpu baseline")
                line["cpu_baseline"] = cpu_baseline(ops, lo, hi, threads)
                progress("done")
            except Exception as exc:                         # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "solves/s", "cores": 0, "kind": "port", "sample": "", "error": repr(exc)}
        print(json.dumps(line), file=real_stdout, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
