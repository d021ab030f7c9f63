# throughput-mode A/B: bash tools/ab_env_tput.sh "<workers>:<VAR=value ...>" ...   ("-" after the colon = defaults)
run() {
  python bench.py --steps 10 --warmup 2 --population 0 --no-cpu-baseline --throughput-workers $1 --no-reference-size 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('throughput_mode') or {}; print('$1 workers, $2 |', round(d['value'],2), 'solves/s isolated; tput', round(t.get('value',0),1), 'ms/step', round(t.get('ms_per_step',0),3))"
}
for rep in 1 2; do
  for arg in "$@"; do
    w=${arg%%:*}; setting=${arg#*:}
    if [ "$setting" = "-" ]; then run $w defaults; else ( export $setting; run $w "$setting" ); fi
  done
done
