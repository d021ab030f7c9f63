"""Summarise rocprofv3 --pmc csv output: per kernel name, mean counter value and duration of the dispatches
(first two dropped). usage: python tools/pmc_summary.py <dir> <COUNTER>"""
import csv, glob, sys, collections
d, counter = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(list)
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows[r["Kernel_Name"][:60]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                                                 int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in rows.items():
    v = sorted(v)[2:] or sorted(v)
    print(f"{counter} {k:60s} n={len(v):3d} mean={sum(x[1] for x in v)/len(v):14.1f} dur_us={sum(x[2] for x in v)/len(v)/1e3:9.1f}")
