"""Developer tool: the headline candidate (C2: 2D Poisson, levels lo..hi, V(1,1) RB-GS) with W candidates in flight on one GPU,
S solves in total — the throughput-mode region of bench.py on its own, for rocprofv3 (kernel trace, PMC passes).
usage: python tools/run_throughput.py [workers=12] [solves=96] [lo=7] [hi=12]
Prints solves/s of the timed region and writes its wall-clock window (ns, CLOCK_MONOTONIC... as rocprofv3 stamps) to stdout."""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
from evostencils_amd import candidates, native, population, problems
from evostencils_amd.code_generation.hip import ProgramGenerator

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 96
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 7
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 12


def make():
    g = ProgramGenerator(problem=problems.poisson_2d(lo, hi), library=native.library(),
                         coarse_cg=os.environ.get("EVOMG_BENCH_COARSE_CG", "single-reduction"))
    g.initialize_code_generation(lo, hi)
    return g


workers = population.ConcurrentEvaluator(make, W)
rb = [("rbgs", 18)]
expr = [candidates.v_cycle(candidates.production_context(g, lo, hi), hi - lo, [rb] * (hi - lo), [rb] * (hi - lo))[0] for g in workers.generators]
for g, e in zip(workers.generators, expr):            # warm-up, one after the other
    g.generate_and_evaluate(e, None, lo, hi, "", evaluation_samples=1)
lock, left = threading.Lock(), [S]


def work(g, e):
    while True:
        with lock:
            if left[0] <= 0:
                return
            left[0] -= 1
        g.generate_and_evaluate(e, None, lo, hi, "", evaluation_samples=1)


def cgroup_cpu():                                       # (usage_usec, nr_throttled, throttled_usec) of this cgroup, zeros where there is none
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            out[k] = int(v)
    except OSError:
        pass
    return out.get("usage_usec", 0), out.get("nr_throttled", 0), out.get("throttled_usec", 0)


threads = [threading.Thread(target=work, args=(g, e)) for g, e in zip(workers.generators, expr)]
cpu0, cg0 = os.times(), cgroup_cpu()
t0n = time.clock_gettime_ns(time.CLOCK_MONOTONIC)   # (rocprofv3 timestamps: CLOCK_BOOTTIME-like; the window is printed in both)
t0 = time.perf_counter()
for t in threads:
    t.start()
for t in threads:
    t.join()
wall = time.perf_counter() - t0
cpu1, cg1 = os.times(), cgroup_cpu()
print("host: %.2f CPU-seconds (user %.2f, system %.2f) in %.3f s of wall = %.1f cores busy; cgroup throttled %d times, %.3f s"
      % (cpu1.user - cpu0.user + cpu1.system - cpu0.system, cpu1.user - cpu0.user, cpu1.system - cpu0.system, wall,
         (cpu1.user - cpu0.user + cpu1.system - cpu0.system) / wall, cg1[1] - cg0[1], (cg1[2] - cg0[2]) / 1e6))
print("throughput: %d workers, %d solves in %.3f s = %.1f solves/s; window_start_ns %d window_ns %d" % (W, S, wall, S / wall, t0n, int(wall * 1e9)))
workers.close()
