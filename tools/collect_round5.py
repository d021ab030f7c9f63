"""Copy what tools/profile_round5.sh left under gpurun_out/r5/ into profiles/ under round-5 names."""
import glob, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(R, "gpurun_out", "r5"), os.path.join(R, "profiles")
names = {"bench_prof": "r05_bench_kernel_stats.csv", "c4_prof": "r05_c4_poisson3d_513_kernel_stats.csv", "c3_prof": "r05_c3_helmholtz_2049_kernel_stats.csv",
         "t9_prof": "r05_t9_poisson2d_513_kernel_stats.csv", "p3d_prof": "r05_poisson3d_65_kernel_stats.csv", "h7_prof": "r05_helmholtz_129_kernel_stats.csv"}
for d, out in names.items():
    hits = glob.glob(os.path.join(src, d, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(max(hits, key=os.path.getmtime), os.path.join(dst, out))      # the newest run (gpurun_out keeps earlier ones)
        print(out)
for f, out in (("bench.json", "r05_bench.json"), ("bench_under_rocprof.json", "r05_bench_under_rocprof.json"), ("configs_plain.txt", "r05_configs_plain.txt")):
    p = os.path.join(src, f)
    if os.path.exists(p):
        if f.endswith(".json"):          # the one JSON line only (whatever else a library printed to the same file)
            lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
            open(os.path.join(dst, out), "w").write((lines[-1] if lines else "") + "\n")
        else:
            shutil.copy(p, os.path.join(dst, out))
        print(out)
