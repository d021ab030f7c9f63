set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
for sm in jacobi rbgs block211 residual; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/r2_pmc_9_${sm}_$c -- python3 $R/tools/pmc_smoother.py 9 12 poisson3d $sm > $O/r2_pmc_9_${sm}_$c.log 2>&1
    python3 $R/tools/pmc_summary.py $O/r2_pmc_9_${sm}_$c $c >> $O/r2_pmc3d_summary.txt
  done
done
echo pmc3d done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/r2_pmc_c2_$c -- python3 $R/tools/run_config.py c2 > $O/r2_pmc_c2_$c.log 2>&1
  python3 $R/tools/pmc_summary.py $O/r2_pmc_c2_$c $c >> $O/r2_pmc_c2_summary.txt
done
echo pmc c2 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2_bench_prof -- python3 $R/bench.py --steps 10 --warmup 2 --population 0 --no-cpu-baseline > $O/r2_bench_under_rocprof.json 2> $O/r2_bench_prof.err
echo bench prof done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2_c4_prof -- python3 $R/tools/run_config.py c4 > $O/r2_c4_prof.log 2>&1
echo c4 prof done
cd $R && python3 bench.py --steps 20 --warmup 5 > $O/r2_bench_b.json 2> $O/r2_bench_b.err
echo bench done
