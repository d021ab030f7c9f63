#!/bin/bash
# Developer tool: rocprofv3 kernel statistics of one named configuration (tools/run_config.py): bash tools/prof_config.sh c3
# Writes gpurun_out/prof_<cfg>/ and prints the twenty kernels with the most time.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cfg=${1:-c3}
O=$R/gpurun_out/prof_$cfg
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/run_config.py $cfg > $O/run.log 2>&1
tail -1 $O/run.log
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:20]:
    print("%-86s calls %6s avg %8.1f us %5.1f %%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
