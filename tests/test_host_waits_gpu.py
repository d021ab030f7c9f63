"""Candidates in flight and the host (csrc/evomg.hip: wait_event / wait_stream): with other evaluations under way the backend's
waits poll and sleep instead of spinning, so that W host threads with a candidate each do not keep W cores busy (a GPU box gives
the process sixteen: profiles/r05_throughput.md). Measured in a child process
as CPU time per wall time of a region in which four threads solve at the same time; results must be those of the sequential
solves, and the waiting threads get their timer slack back (the waits lower it to 1 us while they sleep: ADVICE r5). The
always-spin switch of round 5 (EVOMG_SPIN_WAITS: 3.2 cores busy against 0.9) is retired."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys, threading, time
sys.path.insert(0, os.environ["EVOMG_REPO"])
from evostencils_amd import candidates, native, problems
from evostencils_amd.code_generation.hip import ProgramGenerator
lo, hi, W, S = 6, 11, 4, 60
gens = []
for _ in range(W):
    g = ProgramGenerator(problem=problems.poisson_2d(lo, hi), library=native.library())
    g.initialize_code_generation(lo, hi)
    gens.append(g)
rb = [("rbgs", 18)]
expr = [candidates.v_cycle(candidates.production_context(g, lo, hi), hi - lo, [rb] * (hi - lo), [rb] * (hi - lo))[0] for g in gens]
first = [g.generate_and_evaluate(e, None, lo, hi, "", evaluation_samples=1) for g, e in zip(gens, expr)]      # one after the other
results = [[] for _ in gens]
import ctypes
libc = ctypes.CDLL(None)
slack = [None] * W
def work(k):
    before = libc.prctl(30, 0, 0, 0, 0)                 # PR_GET_TIMERSLACK
    for _ in range(S):
        results[k].append(gens[k].generate_and_evaluate(expr[k], None, lo, hi, "", evaluation_samples=1))
    slack[k] = (before, libc.prctl(30, 0, 0, 0, 0))
threads = [threading.Thread(target=work, args=(k,)) for k in range(W)]
c0, t0 = os.times(), time.perf_counter()
for t in threads: t.start()
for t in threads: t.join()
wall = time.perf_counter() - t0
c1 = os.times()
cpu = (c1.user - c0.user) + (c1.system - c0.system)
same = all(r[1] == first[k][1] and r[2] == first[k][2] for k in range(W) for r in results[k])      # convergence factor, iterations
print("RESULT " + json.dumps({"cores_busy": cpu / wall, "wall": wall, "same": same, "rho": first[0][1], "slack": slack}))
for g in gens: g.close()
"""


def _run():
    env = dict(os.environ, EVOMG_REPO=REPO)
    proc = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    return json.loads([l for l in proc.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])


def test_waits_sleep_while_other_evaluations_are_in_flight():
    sleepy = _run()
    assert sleepy["same"]
    # four threads, each waiting for the device most of the time: spinning they kept (nearly) four cores busy — as far as the
    # box's CPU share let them (round 5: 3.2) —, sleeping the lowering and the launches are all that is left (0.9)
    assert sleepy["cores_busy"] < 2.0, sleepy
    for before, after in sleepy["slack"]:
        assert before == after and before > 1000, sleepy["slack"]          # 50 us by default; never left at the waits' 1 us
