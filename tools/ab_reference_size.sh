# A/B of one environment knob on the bench line's reference-size section (513^2: solves/s one at a time, with twelve in
# flight, solve time). usage (on the GPU box): bash tools/ab_reference_size.sh EVOMG_NO_SMALL2D
knob=${1:-EVOMG_NO_SMALL2D}
show='import json,sys; d=json.loads(sys.stdin.read()); r=d["reference_size"]; print(sys.argv[1], r["value"], r["in_flight"]["value"], r["solve_time_ms"])'
for i in 1 2; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --population 0 --throughput-workers 12 2>/dev/null | python -c "$show" "default     "
  env $knob=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --population 0 --throughput-workers 12 2>/dev/null | python -c "$show" "$knob=1"
done
