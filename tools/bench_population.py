"""Population evaluation benchmark (BASELINE.json configs[4] / BASELINE.md C5): N seeded random candidates on 2D Poisson
2049^2, sharded round-robin over the ranks of a torchrun launch (one rank per GPU), optional K candidates in flight per
GPU, fitness all-gathered over RCCL. usage:
  python tools/bench_population.py [--size 256] [--level 11] [--in-flight 4]
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_population.py"""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--level", type=int, default=11)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--in-flight", type=int, default=1)
    ap.add_argument("--seed", type=int, default=2024)
    args = ap.parse_args()
    import evostencils_amd  # noqa: F401  (first: sets GPU_MAX_HW_QUEUES before the HIP runtime initialises)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from evostencils_amd import candidates, native, population, problems
    from evostencils_amd.code_generation.hip import ProgramGenerator
    context_for = candidates.production_context
    hi = args.level; lo = hi - args.levels + 1

    def make():
        g = ProgramGenerator(problem=problems.poisson_2d(lo, hi), device_id=local_rank, mpi_rank=rank, library=native.library())
        g.initialize_code_generation(lo, hi)
        return g
    gens = [make() for _ in range(max(1, args.in_flight))]
    pop = candidates.population(lambda: context_for(gens[0], lo, hi), hi - lo, 2, args.size, seed=args.seed)
    mine = population.shard_indices(args.size, rank, world)
    # lowering happens inside generate_and_evaluate; warm up each generator once
    for g in gens:
        g.generate_and_evaluate(pop[mine[0]], None, lo, hi, "", evaluation_samples=1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    results = {}
    if len(gens) == 1:
        for i in mine:
            results[i] = gens[0].generate_and_evaluate(pop[i], None, lo, hi, "", evaluation_samples=1)
    else:
        import concurrent.futures as cf, threading
        it = iter(mine); lock = threading.Lock()
        def worker(g):
            while True:
                with lock:
                    i = next(it, None)
                if i is None:
                    return
                results[i] = g.generate_and_evaluate(pop[i], None, lo, hi, "", evaluation_samples=1)
        with cf.ThreadPoolExecutor(len(gens)) as pool:
            list(pool.map(worker, gens))
    rows = population.allgather_fitness([(i, float(r[1]), float(r[0]), float(r[2])) for i, r in results.items()],
                                        args.size, world, local_rank)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        conv = [r for r in rows if r[3] < 1e99]
        print(json.dumps({"config": f"G3P population={args.size}, 2D Poisson {1 << hi}^2, {args.levels} levels",
                          "n_gpus": world, "in_flight_per_gpu": len(gens), "solves_per_s": args.size / elapsed,
                          "wall_s": elapsed, "converged": len(conv), "evaluated": len(rows),
                          "median_time_ms": sorted(r[2] for r in rows)[len(rows) // 2],
                          "best_rho": min(r[1] for r in rows)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
