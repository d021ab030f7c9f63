"""Developer tool: a -D variant of the HIP library under build_exp/ for same-box A/B runs (EVOMG_LIB=build_exp/libevomg_<name>.so).
usage: python tools/build_variant.py <name> [-DMACRO=value ...] [--units impl_real.hip,evomg.hip]
Units not named are taken from the in-tree objects (csrc/obj), so a variant that only touches the real scalar path
recompiles one file."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from evostencils_amd import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    flags = [a for a in sys.argv[2:] if a.startswith("-D")]
    units = B._sources()
    for a in sys.argv[2:]:
        if a.startswith("--units"):
            units = a.split("=", 1)[1].split(",")
    B.build()
    out_dir = os.path.join(REPO, "build_exp", "obj_" + name)
    os.makedirs(out_dir, exist_ok=True)

    def obj(src):
        return os.path.join(out_dir, os.path.splitext(src)[0] + ".o") if src in units else B._obj(src)

    def compile_one(src):
        subprocess.run(["hipcc"] + B.CFLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj(src)], check=True)

    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as pool:
        list(pool.map(compile_one, units))
    lib = os.path.join(REPO, "build_exp", "libevomg_%s.so" % name)
    subprocess.run(["hipcc"] + B.LDFLAGS + ["-o", lib] + [obj(s) for s in B._sources()], check=True)
    print(lib)


if __name__ == "__main__":
    main()
