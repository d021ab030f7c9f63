# Developer tool: SET_ZERO + 3D double sweep without loading the zero field (EVOMG_NO_TWICE_XZERO=1: zero launch + plain double
# sweep) on C4, interleaved, after the 3D peephole tests. usage (GPU box): bash tools/ab_twice_xzero.sh
python -m pytest tests/test_peepholes_gpu.py tests/test_baseline_configs_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x -k "3d or c4 or fuzz" 2>&1 | tail -3
for off in 0 1 0 1; do
  if [ $off = 1 ]; then export EVOMG_NO_TWICE_XZERO=1; else unset EVOMG_NO_TWICE_XZERO; fi
  echo "NO_TWICE_XZERO=$off"; RUN_CONFIG_REPEATS=3 python tools/run_config.py c4 2>&1 | tail -1
done
