"""Copy what tools/profile_round.sh <tag> left under gpurun_out/<tag>/ into profiles/ under that round's names.
usage: python tools/collect_round.py r06"""
import glob, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
src, dst = os.path.join(R, "gpurun_out", TAG), os.path.join(R, "profiles")
names = {"bench_prof": TAG + "_bench_kernel_stats.csv", "c4_prof": TAG + "_c4_poisson3d_513_kernel_stats.csv", "c3_prof": TAG + "_c3_helmholtz_2049_kernel_stats.csv",
         "t9_prof": TAG + "_t9_poisson2d_513_kernel_stats.csv", "p3d_prof": TAG + "_poisson3d_65_kernel_stats.csv", "h7_prof": TAG + "_helmholtz_129_kernel_stats.csv"}
for d, out in names.items():
    hits = glob.glob(os.path.join(src, d, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(max(hits, key=os.path.getmtime), os.path.join(dst, out))      # the newest run (gpurun_out keeps earlier ones)
        print(out)
for f, out in (("bench.json", TAG + "_bench.json"), ("bench_under_rocprof.json", TAG + "_bench_under_rocprof.json"), ("configs_plain.txt", TAG + "_configs_plain.txt")):
    p = os.path.join(src, f)
    if os.path.exists(p):
        if f.endswith(".json"):          # the one JSON line only (whatever else a library printed to the same file)
            lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
            open(os.path.join(dst, out), "w").write((lines[-1] if lines else "") + "\n")
        else:
            shutil.copy(p, os.path.join(dst, out))
        print(out)
