"""The population's exchange step behind the C-ABI (include/evomg.h: evomg_comm_create, evomg_allgather_fitness,
evomg_gather_fitness — RCCL opened by the library itself, no torch) on the GPU box. One GPU = one rank (RCCL refuses two
ranks on one device); what a second rank adds is covered by the world-size-2 gloo tests of the same population functions.
The torch.distributed path is the checker: the same tables through both must agree."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_communicator_moves_the_fitness_table(tmp_path):
    from evostencils_amd import abi, native, population
    lib = native.library()
    assert lib.fn["rccl_version"]() >= 20000
    comm = native.communicator(0, 1, 0, str(tmp_path / "rccl_id"))
    try:
        rows = np.array([[0.0, 0.5, 12.0, 7.0], [1.0, 0.25, 3.5, 9.0], [2.0, 1e100, 1e100, 1e100]])
        assert np.array_equal(comm.allgather(rows), rows)
        assert np.array_equal(comm.gather(rows, 0), rows)
        big = np.arange(4 * 300, dtype=np.float64).reshape(300, 4)          # the device buffers grow on demand
        assert np.array_equal(comm.allgather(big), big)
        # the population functions through the C-ABI communicator: same table as without any collective
        population.use_native_collectives(comm)
        try:
            got = population.evaluate_population(lambda i: (10.0 + i, 0.1 * (i + 1), 5.0), 5)
            assert got == [(i, 0.1 * (i + 1), 10.0 + i, 5.0) for i in range(5)]
            assert population.gather_fitness([(0.5, 12.0, 7.0)], 1) == [(0.5, 12.0, 7.0)]
            with pytest.raises(ZeroDivisionError):                          # a failing evaluator still reaches the collective
                population.evaluate_population(lambda i: 1 / 0, 3)
        finally:
            population.use_native_collectives(None)
    finally:
        comm.close()
    assert not (tmp_path / "rccl_id").exists()                              # rank 0 removes the id file
    with pytest.raises(abi.BackendError):
        native.communicator(1, 2, 0, str(tmp_path / "never_written"), timeout_ms=200)      # bounded wait for the id file
    with pytest.raises(abi.BackendError):
        native.communicator(0, 1, 99, str(tmp_path / "x"))                                 # no such device
    # an id file of ANOTHER launch under the same name (a crashed job's, seconds old): right magic, foreign nonce — a joining
    # rank takes it as absent and its bounded wait says why (round 6, ADVICE r5: validated by a launch nonce, not by its age)
    foreign = tmp_path / "foreign"
    foreign.write_bytes(b"EVOMGID2" + (0x1234567890ABCDEF).to_bytes(8, "little") + bytes(128))
    with pytest.raises(abi.BackendError, match="another launch"):
        native.communicator(1, 2, 0, str(foreign), timeout_ms=200)


CHILD = r"""
import json, os, sys
sys.path.insert(0, os.environ["EVOMG_REPO"])
import evostencils_amd
import numpy as np, torch, torch.distributed as dist
from evostencils_amd import native, population
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    rows = [(i, 0.1 * (i + 1), 10.0 + i, 5.0) for i in range(7)]
    via_torch = population.allgather_fitness(rows, 7, 1, 0)
    comm = native.communicator(0, 1, 0, os.environ["EVOMG_ID_PATH"])
    population.use_native_collectives(comm)
    via_abi = population.allgather_fitness(rows, 7, 1, 0)
    population.use_native_collectives(None)
    comm.close()
    print("RESULT " + json.dumps({"torch": via_torch, "abi": via_abi}))
finally:
    dist.destroy_process_group()
"""


def test_c_abi_collective_agrees_with_the_torch_path(tmp_path):
    """torch's RCCL process group and the library's own communicator in ONE process, the same table through both."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EVOMG_REPO=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0",
               EVOMG_ID_PATH=str(tmp_path / "rccl_id"))
    proc = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = json.loads([l for l in proc.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert out["torch"] == out["abi"] and len(out["abi"]) == 7
