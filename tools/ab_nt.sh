# HISTORICAL (round 3): the libraries under build/ were -DEVOMG_EXPERIMENT_NT_X builds of commit 3250de9's strip kernels; the
# placement they measured (profiles/r03_nt_placement.md) is what evomg_strip2d<.., NT_XY> now does. Kept as the record of the method.
# A/B of the non-temporal placement in the 2D strip kernels on the bench line (same box, interleaved).
# build/libevomg_nt<v>.so = the library built with -DEVOMG_EXPERIMENT_NT_X=<v> (bit 0: x loads, bit 1: y stores non-temporal)
run() {
  python bench.py --steps 20 --warmup 3 --population 0 --no-cpu-baseline --throughput-workers 0 --no-reference-size 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['value'],2), round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'us', round(r['ms_per_launch']*1e3,1), 'b2b', round(r['back_to_back']['frac'],4))"
}
for rep in 1 2 3; do
  unset EVOMG_LIB EVOMG_STRIP_NT; run "A default(NT f)      "
  export EVOMG_LIB=$PWD/build/libevomg_ntx.so EVOMG_STRIP_NT=0; run "C NT x,y; f temporal "
  export EVOMG_LIB=$PWD/build/libevomg_nt1.so EVOMG_STRIP_NT=0; run "E NT x; y,f temporal "
  export EVOMG_LIB=$PWD/build/libevomg_nt2.so EVOMG_STRIP_NT=0; run "F NT y; x,f temporal "
done
