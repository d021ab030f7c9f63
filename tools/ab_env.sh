# A/B of environment settings on the bench line (same box, interleaved): bash tools/ab_env.sh "A=1 B=2" "C=3" ...
# each argument is one setting (space-separated VAR=value pairs; "-" = defaults)
run() {
  python bench.py --steps 20 --warmup 3 --population 0 --no-cpu-baseline --throughput-workers ${TW:-0} --no-reference-size 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d.get('throughput_mode') or {}; print('$1 |', round(d['value'],2), 'solves/s', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],4), 'us', round(r['ms_per_launch']*1e3,1), 'b2b', round(r['back_to_back']['frac'],4), 'tput', round(t.get('value',0),1))"
}
for rep in 1 2 3; do
  for setting in "$@"; do
    if [ "$setting" = "-" ]; then run "defaults"; else ( export $setting; run "$setting" ); fi
  done
done
