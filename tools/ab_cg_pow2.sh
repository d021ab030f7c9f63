# Developer tool: the 129^2 register CG with and without the power-of-two form (EVOMG_NO_POW2_FMA=1 also switches the
# smoothers' forms off), C2 solve time and the coarse-solve parity tests. usage (GPU box): bash tools/ab_cg_pow2.sh
python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py tests/test_cross_fusion_gpu.py -m gpu -q -x -k "coarse or cg or c2 or 2d or cross" 2>&1 | tail -3
for off in 0 1 0 1; do
  if [ $off = 1 ]; then export EVOMG_NO_POW2_FMA=1; else unset EVOMG_NO_POW2_FMA; fi
  echo "NO_POW2_FMA=$off"; RUN_CONFIG_REPEATS=5 python tools/run_config.py c2 2>&1 | tail -2
done
