# Developer tool: the power-of-two FMA forms of the 3D tile kernels (EVOMG_NO_POW2_FMA switches them off) on C4, interleaved,
# then the 3D parity tests and back-to-back timings of the 513^3 kernels. usage (GPU box): bash tools/ab_pow2_3d.sh
for off in 0 1 0 1; do
  if [ $off = 1 ]; then export EVOMG_NO_POW2_FMA=1; else unset EVOMG_NO_POW2_FMA; fi
  echo "NO_POW2_FMA=$off"; RUN_CONFIG_REPEATS=3 python tools/run_config.py c4 2>&1 | tail -2
done
unset EVOMG_NO_POW2_FMA
python -m pytest tests/test_full_size_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_peepholes_gpu.py tests/test_random_programs_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q -x -k "3d or 3D or poisson3 or c4 or random_programs or peephole" 2>&1 | tail -4
for off in 0 1; do
if [ $off = 1 ]; then export EVOMG_NO_POW2_FMA=1; else unset EVOMG_NO_POW2_FMA; fi
python - <<PY
import sys
sys.path[:0]=["/root/repo","/root/repo/tests"]
from evostencils_amd import abi, native, problems
pd = problems.poisson_3d(5, 9)
h = native.create(pd)
for name, op in (("jacobi", abi.make_op(abi.OP_SMOOTH, 9, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1,1,1), 0.8)),
                 ("rbgs", abi.make_op(abi.OP_SMOOTH, 9, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, (1,1,1), 1.0)),
                 ("block211", abi.make_op(abi.OP_SMOOTH, 9, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (2,1,1), 0.8)),
                 ("residual", abi.make_op(abi.OP_RESIDUAL, 9, abi.F_RES, abi.F_SOL, abi.F_RHS))):
    b = sorted((h.bench_op(op, warmup=3, reps=20) for _ in range(3)), key=lambda b: b.ms)[1]
    print("NO_POW2_FMA=$off", name, "%.1f us  frac %.3f" % (1e3*b.ms, b.gbps/8000))
PY
done
