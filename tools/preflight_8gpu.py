#!/usr/bin/env python3
"""Preflight for a multi-GPU run of bench.py / an optimisation: checks, per rank and BEFORE the first HIP call of the
process, everything the one-process-per-GPU path assumes (evostencils_amd/preflight.py lists the checks) — run it under the
same launcher as the real job:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/preflight_8gpu.py [--collective]

(or plainly `python tools/preflight_8gpu.py` for the single-process checks). Prints one JSON object per rank and exits
non-zero on the first failed check. bench.py runs the same checks at start-up (`config.collectives.preflight`)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    queues_before = os.environ.get("GPU_MAX_HW_QUEUES")
    torch_before = "torch" in sys.modules
    from evostencils_amd import preflight               # importing the package sets GPU_MAX_HW_QUEUES if the user did not
    rep = preflight.report(collective="--collective" in sys.argv, torch_imported_before=torch_before)
    rep["GPU_MAX_HW_QUEUES_set_by"] = "user" if queues_before else "evostencils_amd"
    print(json.dumps(rep), flush=True)
    return 0 if rep["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
