# A/B of one environment knob on the bench line (same box, interleaved): bash tools/ab_bench.sh EVOMG_NO_CACHE_WARMUP
knob=$1
for v in 0 1 0 1 0 1; do
  if [ $v = 1 ]; then export $knob=1; else unset $knob; fi
  python bench.py --steps 20 --warmup 3 --population 0 --no-cpu-baseline --throughput-workers 0 --no-reference-size 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$knob=$v', round(d['value'],2), round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'us', round(r['ms_per_launch']*1e3,1), 'b2b', round(r['back_to_back']['frac'],4))"
done
