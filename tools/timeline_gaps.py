"""Busy time, gaps and the longest kernels of a rocprofv3 kernel trace (csv): where a launch-bound solve spends its time.
usage: python tools/timeline_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# keep the second half (steady state)
rows = rows[len(rows) // 2:]
span = rows[-1][1] - rows[0][0]
busy, cur_end, gaps = 0, rows[0][0], []
for s, e, _ in rows:
    if s > cur_end:
        gaps.append(s - cur_end)
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
print("kernels %d, span %.2f ms, union busy %.2f ms (%.0f %%), mean per kernel: span %.2f us, duration %.2f us" % (
    len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, span / 1e3 / len(rows), sum(e - s for s, e, _ in rows) / 1e3 / len(rows)))
gaps.sort()
print("gaps: n %d, median %.2f us, p90 %.2f us, sum %.2f ms; gaps > 10 us: %d (sum %.2f ms)" % (
    len(gaps), gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, sum(gaps) / 1e6, sum(1 for g in gaps if g > 10000), sum(g for g in gaps if g > 10000) / 1e6))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    k = n[:70]
    agg[k][0] += 1
    agg[k][1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print("%6d x %6.2f us  %s" % (c, t / 1e3 / c, k))
