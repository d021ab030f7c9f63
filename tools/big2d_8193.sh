python tools/bench_ops.py poisson2d 13 10 2>&1 | tail -12
python - <<'PY'
import sys, dataclasses
sys.path[:0]=["/root/repo","/root/repo/tests"]
from evostencils_amd import abi, native, problems
from test_full_size_gpu import vcycle
lo, hi = 8, 13
pd = problems.poisson_2d(lo, hi)
h = native.create(pd)
ops = vcycle(lo, hi, 1, 1, lambda L: abi.make_op(abi.OP_SMOOTH, L, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1, 1, 1), 1.0))
h.evaluate(ops, 1, 0, 0)
h.profile_begin(abi.OP_SMOOTH, hi)
r = h.evaluate(ops, 1, 0, 0)
b = h.profile_end()
print("8193^2 V(1,1) RB-GS, levels 13..8: %.2f ms per solve, %d iterations; finest-level smoothing launches inside the solve: %d x %.1f us, %.0f MB algorithmic each = %.3f of 8 TB/s [%s]" % (r.time_ms, r.n_iterations, b.launches, 1e3*b.ms, b.bytes/1e6, b.gbps/8000, h.kernel_name(ops[0])))
PY
