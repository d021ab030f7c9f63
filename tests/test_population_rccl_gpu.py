"""The fitness collectives on the `nccl` backend (= RCCL on ROCm) on the GPU box. One GPU is all a gpurun box has, so the
process group has ONE rank: RCCL is loaded, its communicator is built on cuda:0 and its kernels move the module's float64
tables on the device — everything of the N > 1 path except a second GPU (the sharding logic itself is covered by the
world-size-2 gloo tests). Runs in a child process so that the process group does not leak into the test session."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, os.environ["EVOMG_REPO"])
import evostencils_amd                      # before torch: hardware-queue setting
import torch, torch.distributed as dist
from evostencils_amd import population
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    out = population.collective_selftest(0)
    rows = population.evaluate_population(lambda i: (10.0 + i, 0.1 * (i + 1), 5.0), 3)
    print("RESULT " + json.dumps({"selftest": out, "rows": rows}))
finally:
    dist.destroy_process_group()
"""


def test_fitness_collectives_run_on_rccl():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EVOMG_REPO=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = [l for l in proc.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    st = out["selftest"]
    assert st["backend"] == "nccl" and st["device"].startswith("cuda")
    assert st["all_gather"] == [[[0.0, 0.5, 12.0, 7.0]]] and st["gather"] == [[[0.0, 0.5, 12.0, 7.0]]]
    assert st["all_reduce"][0] == [1.0, 0.25, 3.0, 9.0] and st["all_reduce"][1] == [0.0] * 4
    assert [r[0] for r in out["rows"]] == [0, 1, 2]
