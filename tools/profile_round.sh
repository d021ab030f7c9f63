#!/bin/bash
# Profile refresh of a round (run on the GPU box: bash tools/profile_round.sh r06). Writes under gpurun_out/<tag>/; copy what
# is to be judged into profiles/ afterwards (python tools/collect_round.py r06).
set -e
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
# kernel-trace statistics of the bench command and of each named configuration
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python3 $R/bench.py --steps 10 --warmup 2 --population 0 --no-cpu-baseline --throughput-workers 0 --no-reference-size --no-configs > $O/bench_under_rocprof.json 2> $O/bench_prof.err
echo bench prof done
for cfg in c4 c3 t9 h7 p3d; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${cfg}_prof -- python3 $R/tools/run_config.py $cfg > $O/${cfg}_prof.log 2>&1
  echo $cfg done
done
# the configurations one at a time WITHOUT the profiler
for cfg in c2 c3 c4 c5 t9 t8 h7 p3d ela fas; do RUN_CONFIG_REPEATS=3 python3 $R/tools/run_config.py $cfg 2>&1 | tail -2; done > $O/configs_plain.txt
echo plain configs done
cd $R && python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo bench done
