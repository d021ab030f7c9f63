#!/usr/bin/env python3
"""Preflight for a multi-GPU run of bench.py / an optimisation: checks, per rank and BEFORE the first HIP call of the
process, everything the one-process-per-GPU path assumes — run it under the same launcher as the real job:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/preflight_8gpu.py

(or plainly `python tools/preflight_8gpu.py` for the single-process checks). Prints one JSON object per rank and exits
non-zero on the first failed check. Checks: GPU_MAX_HW_QUEUES is in the environment before the runtime initialises
(evostencils_amd/__init__.py); HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: RCCL across processes needs it on this pool);
evomg_device_count() >= ranks on the node; LOCAL_RANK -> device as ProgramGenerator resolves it, one distinct device per
local rank; librccl opens (evomg_rccl_version); with --collective: the C-ABI communicator is built across all ranks and one
evomg_allgather_fitness round trip returns every rank's row."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    report = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
              "world": int(os.environ.get("WORLD_SIZE", "1")), "local_world": int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))}
    failures = []
    queues_before = os.environ.get("GPU_MAX_HW_QUEUES")
    import evostencils_amd                                  # noqa: F401 — sets GPU_MAX_HW_QUEUES if the user did not
    report["GPU_MAX_HW_QUEUES"] = os.environ.get("GPU_MAX_HW_QUEUES")
    report["GPU_MAX_HW_QUEUES_set_by"] = "user" if queues_before else "evostencils_amd"
    if "torch" in sys.modules:
        failures.append("torch was imported before evostencils_amd: the hardware-queue setting may be too late")
    report["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    if report["world"] > 1 and report["HSA_ENABLE_IPC_MODE_LEGACY"] != "0":
        failures.append("HSA_ENABLE_IPC_MODE_LEGACY must be 0 for multi-process RCCL on this pool")
    from evostencils_amd import native
    from evostencils_amd.code_generation.hip import ProgramGenerator
    lib = native.library()
    report["rccl_version"] = lib.fn["rccl_version"]()       # dlopen only: nothing touches the GPU yet
    if report["rccl_version"] <= 0:
        failures.append("librccl could not be opened")
    count = lib.fn["device_count"]()                        # the first HIP call of this process
    report["device_count"] = count
    if count < report["local_world"]:
        failures.append(f"{report['local_world']} ranks on this node but {count} visible GPUs")
    device = ProgramGenerator._default_device(report["rank"], lib)
    report["device"] = device
    if count >= report["local_world"] and device != report["local_rank"] % max(count, 1):
        failures.append(f"local rank {report['local_rank']} resolved to device {device}")
    if "--collective" in sys.argv and not failures:
        import numpy as np
        port = os.environ.get("MASTER_PORT", "0")
        comm = native.communicator(report["rank"], report["world"], device, f"/dev/shm/evomg_preflight_{port}", timeout_ms=30000)
        table = comm.allgather(np.array([[report["rank"], device, os.getpid(), 1.0]]))
        comm.close()
        report["allgather"] = table.tolist()
        if [int(r[0]) for r in table] != list(range(report["world"])):
            failures.append("evomg_allgather_fitness returned the wrong rows")
        if len({int(r[1]) for r in table}) != min(report["world"], count):
            failures.append("ranks share devices")
    report["ok"] = not failures
    report["failures"] = failures
    print(json.dumps(report), flush=True)
    return 0 if not failures else 1


if __name__ == "__main__":
    sys.exit(main())
