"""Preflight of the one-process-per-GPU path: what a rank can check about its environment before (and with) its first HIP
call — used by tools/preflight_8gpu.py under the real launcher and by bench.py at start-up (the report goes on the bench
line, `config.collectives.preflight`).

Checks: GPU_MAX_HW_QUEUES in the environment before the runtime initialises (evostencils_amd/__init__.py);
HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: RCCL across processes needs it on this pool); evomg_device_count() >= ranks on the
node; LOCAL_RANK -> device as ProgramGenerator resolves it, one distinct device per local rank; librccl opens
(evomg_rccl_version); with collective=True: the C-ABI communicator is built across all ranks and one evomg_allgather_fitness
round trip returns every rank's row (optimization/program.py:582 is one such exchange per generation); round 6: the host
budget — usable cores (affinity and cgroup quota) against two per local rank (`host_budget`: a warning, not a failure)."""
import os
import sys


def id_path(tag="preflight"):
    """A rendezvous file name unique per launch: the ranks of one torchrun share MASTER_PORT and their parent (the launcher's
    agent); a rerun after a crash gets another name, so no rank can pick up a stale id."""
    return f"/dev/shm/evomg_{tag}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"


def usable_cores():
    """CPUs this process (and, under one launcher, its sibling ranks together) may actually use: the affinity mask, cut down
    to the cgroup's quota where there is one — a GPU box shows all 256 hardware threads to a container throttled to 16."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            text = open(path).read().split()
            if path.endswith("cpu.max"):
                if text[0] != "max":
                    n = min(n, max(1, int(int(text[0]) / int(text[1]))))
            else:
                quota = int(text[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def host_budget(local_world, workers_per_rank=16, cores=None):
    """Host side of `local_world` ranks with `workers_per_rank` candidates in flight each: every worker is a host
    thread that sleeps in its waits while other evaluations are under way (1-2 cores busy per rank, profiles/r05_throughput.md)
    and spins only when its evaluation is alone on the device (one core). Budget: two cores per local rank; below that the
    waits of one rank delay the launches of another and the projection of bench.py's `projected_8gpu` (zero host interference)
    does not hold. Returns a dict with `ok` False and a warning then — a warning, not a failure: the run is still correct."""
    cores = usable_cores() if cores is None else cores
    need = 2 * max(1, local_world)
    rep = {"usable_cores": cores, "local_ranks": local_world, "worker_threads": local_world * workers_per_rank,
           "cores_per_rank": cores / max(1, local_world), "needed_cores": need, "ok": cores >= need}
    if not rep["ok"]:
        rep["warning"] = (f"{cores} usable cores for {local_world} ranks x {workers_per_rank} worker threads: fewer than two per "
                          f"rank; expect host-side interference between ranks (set fewer candidates in flight, or widen the cgroup)")
    return rep


def report(collective=False, torch_imported_before=None):
    rep = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
           "world": int(os.environ.get("WORLD_SIZE", "1")),
           "local_world": int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))}
    failures = []
    rep["GPU_MAX_HW_QUEUES"] = os.environ.get("GPU_MAX_HW_QUEUES")
    if torch_imported_before is None:
        torch_imported_before = False
    if torch_imported_before:
        failures.append("torch was imported before evostencils_amd: the hardware-queue setting may be too late")
    rep["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    if rep["world"] > 1 and rep["HSA_ENABLE_IPC_MODE_LEGACY"] != "0":
        failures.append("HSA_ENABLE_IPC_MODE_LEGACY must be 0 for multi-process RCCL on this pool")
    from evostencils_amd import native
    from evostencils_amd.code_generation.hip import ProgramGenerator
    lib = native.library()
    rep["rccl_version"] = lib.fn["rccl_version"]()       # dlopen only: nothing touches the GPU yet
    if rep["rccl_version"] <= 0:
        failures.append("librccl could not be opened")
    count = lib.fn["device_count"]()                     # possibly the first HIP call of this process
    rep["device_count"] = count
    if count < rep["local_world"]:
        failures.append(f"{rep['local_world']} ranks on this node but {count} visible GPUs")
    device = ProgramGenerator._default_device(rep["rank"], lib)
    rep["device"] = device
    if count >= rep["local_world"] and device != rep["local_rank"] % max(count, 1):
        failures.append(f"local rank {rep['local_rank']} resolved to device {device}")
    if collective and not failures:
        import numpy as np
        comm = native.communicator(rep["rank"], rep["world"], device, id_path(), timeout_ms=30000)
        table = comm.allgather(np.array([[rep["rank"], device, os.getpid(), 1.0]]))
        comm.close()
        rep["allgather"] = table.tolist()
        if [int(r[0]) for r in table] != list(range(rep["world"])):
            failures.append("evomg_allgather_fitness returned the wrong rows")
        if len({int(r[1]) for r in table}) != min(rep["world"], count):
            failures.append("ranks share devices")
    rep["host_budget"] = host_budget(rep["local_world"])          # a warning inside, never a failure
    rep["ok"] = not failures
    rep["failures"] = failures
    return rep
