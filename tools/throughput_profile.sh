#!/bin/bash
# Throughput mode under the profiler (GPU box): kernel trace of W (default 16) candidates in flight, then FETCH_SIZE / WRITE_SIZE per
# dispatch of ONE candidate's solve (counter passes serialise the kernels: bytes per solve are taken there, the rate from the
# unprofiled run). usage: bash tools/throughput_profile.sh [tag]; results under gpurun_out/<tag>_*
set -e
tag=${1:-r05_tput}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 $R/tools/run_throughput.py ${W:-16} 128 > $O/${tag}_plain.txt 2>&1
cat $O/${tag}_plain.txt | tail -1
rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_trace -- python3 $R/tools/run_throughput.py ${W:-16} 128 > $O/${tag}_trace.log 2>&1
window=$(grep -o "window_ns [0-9]*" $O/${tag}_trace.log | tail -1 | cut -d" " -f2)
python3 $R/tools/throughput_trace.py $O/${tag}_trace $window > $O/${tag}_trace.txt
cat $O/${tag}_trace.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${tag}_$c -- python3 $R/tools/run_throughput.py 1 4 > $O/${tag}_$c.log 2>&1
done
python3 $R/tools/pmc_table.py $O/${tag}_FETCH_SIZE $O/${tag}_WRITE_SIZE > $O/${tag}_pmc.txt
cat $O/${tag}_pmc.txt
rm -rf $O/${tag}_trace/*/*.db $O/${tag}_FETCH_SIZE/*/*.db $O/${tag}_WRITE_SIZE/*/*.db 2>/dev/null || true
