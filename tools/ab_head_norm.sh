# Developer tool: the 3D loop norm riding on the coming cycle's first double sweep (EVOMG_NO_HEAD_NORM=1: a norm pass of its
# own, as before) on C4, interleaved, after the 3D parity tests. usage (GPU box): bash tools/ab_head_norm.sh
python -m pytest tests/test_full_size_gpu.py tests/test_parity_gpu.py tests/test_baseline_configs_gpu.py tests/test_peepholes_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x -k "3d or 3D or poisson3 or c4 or peephole or fuzz" 2>&1 | tail -3
for off in 0 1 0 1; do
  if [ $off = 1 ]; then export EVOMG_NO_HEAD_NORM=1; else unset EVOMG_NO_HEAD_NORM; fi
  echo "NO_HEAD_NORM=$off"; RUN_CONFIG_REPEATS=3 python tools/run_config.py c4 2>&1 | tail -2
done
