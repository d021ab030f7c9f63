# Round-3 profile refresh (run on the GPU box: bash tools/profile_round3.sh). Writes under gpurun_out/r3/; copy what is to
# be judged into profiles/ afterwards (tools/collect_round3.py).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
# kernel-trace statistics of the bench command and of each named configuration
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python3 $R/bench.py --steps 10 --warmup 2 --population 0 --no-cpu-baseline --throughput-workers 0 --no-reference-size > $O/bench_under_rocprof.json 2> $O/bench_prof.err
echo bench prof done
for cfg in c4 c3 t9 ela p3d h7 fas; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${cfg}_prof -- python3 $R/tools/run_config.py $cfg > $O/${cfg}_prof.log 2>&1
  echo $cfg done
done
# traffic of the C2 kernels (separate PMC passes)
bash $R/tools/pmc_sq.sh c2 r3/c2 > /dev/null
echo pmc tables done
cd $R && python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo bench done
