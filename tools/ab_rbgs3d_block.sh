# Developer tool: both colours of a 3D red-black sweep in one launch below the tile threshold (EVOMG_NO_RBGS3D_BLOCK=1: two
# thread-per-node launches, as before) — the 3D parity tests, then the reference's 3D example at its own size (65^3, V(2,1)).
python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_peepholes_gpu.py tests/test_random_programs_gpu.py tests/test_grammar_individuals_gpu.py tests/test_problem_files_gpu.py -m gpu -q -x -k "3d or 3D or poisson3 or random_programs or fuzz or individuals or harmonic" 2>&1 | tail -3
for off in 0 1 0 1; do
  if [ $off = 1 ]; then export EVOMG_NO_RBGS3D_BLOCK=1; else unset EVOMG_NO_RBGS3D_BLOCK; fi
  echo "NO_RBGS3D_BLOCK=$off"; RUN_CONFIG_REPEATS=7 python tools/run_config.py p3d 2>&1 | tail -2
done
