# Developer tool: residual + restriction in one launch on small 3D levels (EVOMG_NO_RR3D_BLOCK=1: two launches) on the reference's
# 3D example (65^3), interleaved, after the small-3D tests. usage (GPU box): bash tools/ab_rr3d_block.sh
python -m pytest tests/test_peepholes_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_grammar_individuals_gpu.py -m gpu -q -x -k "3d or 3D or poisson3" 2>&1 | tail -3
for off in 0 1 0 1; do
  if [ $off = 1 ]; then export EVOMG_NO_RR3D_BLOCK=1; else unset EVOMG_NO_RR3D_BLOCK; fi
  echo "NO_RR3D_BLOCK=$off"; RUN_CONFIG_REPEATS=9 python tools/run_config.py p3d 2>&1 | tail -1
done
