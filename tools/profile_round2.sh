# Round-2 profile refresh (run on the GPU box: bash tools/profile_round2.sh). Writes under gpurun_out/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -f $O/r2_pmc3d_summary.txt $O/r2_pmc_c2_summary.txt
# 3D single-kernel traffic at 513^3
for sm in jacobi rbgs block211 residual; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/r2_pmc_9_${sm}_$c -- python3 $R/tools/pmc_smoother.py 9 12 poisson3d $sm > $O/r2_pmc_9_${sm}_$c.log 2>&1
    python3 $R/tools/pmc_summary.py $O/r2_pmc_9_${sm}_$c $c >> $O/r2_pmc3d_summary.txt
  done
done
echo pmc3d done
# C2 / C4: SQ counters + traffic per kernel
bash $R/tools/pmc_sq.sh c2 r2f_c2 > /dev/null
bash $R/tools/pmc_sq.sh c4 r2f_c4 > /dev/null
echo pmc tables done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2_bench_prof -- python3 $R/bench.py --steps 10 --warmup 2 --population 0 --no-cpu-baseline --throughput-workers 0 --no-reference-size > $O/r2_bench_under_rocprof.json 2> $O/r2_bench_prof.err
echo bench prof done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2_c4_prof -- python3 $R/tools/run_config.py c4 > $O/r2_c4_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2_t9_prof -- python3 $R/tools/run_config.py t9 > $O/r2_t9_prof.log 2>&1
echo c4 t9 prof done
cd $R && python3 tools/bench_ops.py poisson3d 9 20 > $O/r2_ops9.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $O/r2_bench_b.json 2> $O/r2_bench_b.err
echo bench done
