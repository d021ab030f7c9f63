"""Per kernel (name + grid size): mean of every counter found in the given rocprofv3 --pmc output directories.
usage: python tools/pmc_table.py <dir> [<dir> ...]"""
import csv, glob, sys, collections
tab = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            key = (r["Kernel_Name"][:52], r["Grid_Size"])
            tab[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
names = sorted({c for v in tab.values() for c in v})
print("kernel".ljust(52), "grid".rjust(9), "us".rjust(8), *(c[-14:].rjust(15) for c in names))
for key, v in sorted(tab.items(), key=lambda kv: -sum(dur[kv[0]])):
    if sum(dur[key]) < 0.01 * sum(map(sum, dur.values())):
        continue
    print(key[0].ljust(52), key[1].rjust(9), ("%.1f" % (sum(dur[key]) / len(dur[key]) / 1e3)).rjust(8),
          *(("%.4g" % (sum(v[c]) / len(v[c])) if c in v else "-").rjust(15) for c in names))
