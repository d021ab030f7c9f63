"""Developer tool: soak of the multi-workgroup coarse-grid CG kernels (csrc/evomg_cg1.h) — their rendezvous is the one place of
the backend where workgroups wait for each other. (1) `solves` solves of 129^2 and of 33^3 one after the other on one handle,
every solution compared bit for bit with the first (sequence numbers, parity buffers, the XCD-local exchange); (2) `workers`
threads, own handle each, running such solves at the same time (each solve on one XCD, dealt round-robin; co-residency under
load) — again every solution bit-identical to the thread's first. Prints failures (mismatches, device errors).
usage: python tools/soak_cg1.py [solves=2000] [workers=16] [rounds=60]"""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
from evostencils_amd import abi, native, problems

solves = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 60


def setup(dim, lo, seed):
    pd = problems.poisson_2d(lo, lo + 1) if dim == 2 else problems.poisson_3d(lo, lo + 1)
    h = native.create(pd)
    n = (1 << lo) + 1
    rhs = np.zeros((n,) * dim)
    rhs[(slice(1, -1),) * dim] = np.random.default_rng(seed).standard_normal((n - 2,) * dim)
    h.set_field(lo, abi.F_CGS_RHS, rhs)
    ops = [abi.make_op(abi.OP_SET_ZERO, lo, abi.F_CORR), abi.make_op(abi.OP_COARSE_SOLVE, lo)]
    return h, ops, lo


bad = 0
t0 = time.time()
for dim, lo in ((2, 7), (3, 5)):
    h, ops, lo = setup(dim, lo, 1)
    h.run_ops(ops)
    first = h.get_field(lo, abi.F_CORR)
    batch = 50
    for k in range(0, solves, batch):
        h.run_ops(ops * batch)
        if not np.array_equal(h.get_field(lo, abi.F_CORR), first):
            bad += 1
            print("mismatch after", k + batch, "solves, dim", dim)
    h.close()
    print(f"sequential: {solves} solves of {'129^2' if dim == 2 else '33^3'} bit-identical: {bad == 0}  ({time.time() - t0:.1f} s)", flush=True)

errors = []


def worker(i):
    try:
        h, ops, lo = setup(2 if i % 4 else 3, 7 if i % 4 else 5, 100 + i)
        h.run_ops(ops)
        first = h.get_field(lo, abi.F_CORR)
        for r in range(rounds):
            h.run_ops(ops * 4)
            if not np.array_equal(h.get_field(lo, abi.F_CORR), first):
                errors.append(("mismatch", i, r))
        h.close()
    except Exception as exc:      # noqa: BLE001
        errors.append(("error", i, repr(exc)))


threads = [threading.Thread(target=worker, args=(i,)) for i in range(workers)]
t1 = time.time()
for t in threads:
    t.start()
for t in threads:
    t.join()
print(f"in flight: {workers} workers x {rounds} rounds x 4 solves in {time.time() - t1:.1f} s, failures: {len(errors)}")
for e in errors[:10]:
    print("  ", e)
print("total failures:", bad + len(errors))
