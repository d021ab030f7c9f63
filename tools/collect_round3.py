"""Copy what tools/profile_round3.sh left under gpurun_out/r3/ into profiles/ under round-3 names."""
import glob, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(R, "gpurun_out", "r3"), os.path.join(R, "profiles")
names = {"bench_prof": "r03_bench_kernel_stats.csv", "c4_prof": "r03_c4_poisson3d_513_kernel_stats.csv", "c3_prof": "r03_c3_helmholtz_2049_kernel_stats.csv",
         "t9_prof": "r03_t9_poisson2d_513_kernel_stats.csv", "ela_prof": "r03_elasticity_257_kernel_stats.csv", "p3d_prof": "r03_poisson3d_65_kernel_stats.csv",
         "h7_prof": "r03_helmholtz_129_kernel_stats.csv", "fas_prof": "r03_fas_1025_kernel_stats.csv"}
for d, out in names.items():
    hits = glob.glob(os.path.join(src, d, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(dst, out))
        print(out)
for f, out in (("bench.json", "r03_bench.json"), ("bench_under_rocprof.json", "r03_bench_under_rocprof.json"), ("c2_table.txt", "r03_pmc_tables.txt"),
               ("gpu_tests.log", "r03_gpu_tests.log")):
    p = os.path.join(src, f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, out))
        print(out)
