set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( bash tools/throughput_profile.sh r06_tput > gpurun_out/r06/tput_profile.log 2>&1 ) || echo tput profile failed
( bash tools/pmc_sq.sh c4 r06_pmc_c4 > gpurun_out/r06/pmc_c4.log 2>&1 ) || echo pmc c4 failed
( bash tools/pmc_sq.sh c2 r06_pmc_c2 > gpurun_out/r06/pmc_c2.log 2>&1 ) || echo pmc c2 failed
cd $GRAFT_REPO_ROOT
python tools/host_overhead.py 5 9 > gpurun_out/r06/host_overhead_5_9.txt 2>&1
python tools/host_overhead.py 7 12 > gpurun_out/r06/host_overhead_7_12.txt 2>&1
for rows in 0 14 18 22 28 36; do echo "EVOMG_STRIP_ROWS=$rows"; EVOMG_STRIP_ROWS=$rows RUN_CONFIG_REPEATS=3 python tools/run_config.py c3 2>&1 | tail -2; done > gpurun_out/r06/c3_rows.txt 2>&1
rm -rf gpurun_out/r06_*/*/*.db gpurun_out/r06_*_sq/*/*.db 2>/dev/null
du -sh gpurun_out | tail -1
