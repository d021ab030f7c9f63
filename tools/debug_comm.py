"""Developer tool: evomg_comm_create AFTER other GPU work in the same process (NCCL_DEBUG output on stderr)."""
import os, sys
os.environ.setdefault("NCCL_DEBUG", "INFO")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import numpy as np
from evostencils_amd import abi, native, problems
from test_full_size_gpu import vcycle
mode = sys.argv[1] if len(sys.argv) > 1 else "after"
pd = problems.poisson_2d(5, 9)
ops = vcycle(5, 9, 1, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.0))
if mode == "after":
    hs = [native.create(pd) for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3)]
    for h in hs:
        h.evaluate(ops, 1, 0, 0)
    if len(sys.argv) > 3:
        for h in hs:
            h.close()
comm = native.communicator(0, 1, 0, "/tmp/evomg_debug_id")
print("allgather", comm.allgather(np.array([[0.0, 1.0, 2.0, 3.0]])).tolist())
comm.close()
print("ok", mode)
