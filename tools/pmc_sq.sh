# SQ occupancy / stall counters of one configuration's kernels (one pass), then FETCH_SIZE and WRITE_SIZE (own passes)
# usage (on the GPU box): bash tools/pmc_sq.sh c4 tag
set -e
cfg=${1:-c4}; tag=${2:-sq}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/${tag}_sq -- python3 $R/tools/run_config.py $cfg > $O/${tag}_sq.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${tag}_$c -- python3 $R/tools/run_config.py $cfg > $O/${tag}_$c.log 2>&1
done
python3 $R/tools/pmc_table.py $O/${tag}_sq $O/${tag}_FETCH_SIZE $O/${tag}_WRITE_SIZE > $O/${tag}_table.txt
cat $O/${tag}_table.txt
