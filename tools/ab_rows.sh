#!/bin/bash
# A/B of the strip chunk height on the bench (developer tool)
for r in "$@"; do
  EVOMG_STRIP_ROWS=$r python bench.py --steps 10 --warmup 2 --no-cpu-baseline --throughput-workers 0 > gpurun_out/ab_rows_$r.json 2>/dev/null
  python - "$r" <<'PY'
import json, sys
r = sys.argv[1]
d = json.load(open(f"gpurun_out/ab_rows_{r}.json")); q = d["roofline"]
print("rows", r, "solves/s %.2f" % d["value"], "solve ms %.2f" % d["config"]["solve_time_ms"], "frac %.3f" % q["frac"], "ms/launch %.4f" % q["ms_per_launch"], "b2b %.3f" % q["back_to_back"]["frac"])
PY
done
