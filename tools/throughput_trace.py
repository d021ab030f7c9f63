"""Developer tool: what a rocprofv3 --kernel-trace of tools/run_throughput.py says about candidates in flight: the timed region
is taken as the last `solves` x (launches per solve) dispatches; per kernel family: launches, mean duration, share of the
summed kernel time; how many finest-level (bandwidth-carrying) launches run at once, as a share of the wall time.
usage: python tools/throughput_trace.py <dir with *_kernel_trace.csv> [window_ns_from_the_end]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], int(r.get("Grid_Size") or r["Grid_Size_X"])) for r in csv.DictReader(open(f))]
rows.sort()
end = max(r[1] for r in rows)
window = int(sys.argv[2]) if len(sys.argv) > 2 else (end - rows[0][0]) // 2
rows = [r for r in rows if r[0] >= end - window]
span = rows[-1][1] - rows[0][0]
print("window %.1f ms, %d dispatches on %d queues" % (span / 1e6, len(rows), len({r[3] for r in rows})))
fam = collections.defaultdict(lambda: [0, 0])
for s, e, n, q, g in rows:
    k = n.split("<")[0].split("(")[0][:40]
    if "strip2d_cross" in n or ("strip2d" in n and g >= 400 * 256): k += " [finest]"
    fam[k][0] += 1
    fam[k][1] += e - s
tot = sum(v[1] for v in fam.values())
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-52s n=%6d mean %8.1f us  %5.1f %% of kernel time" % (k, c, t / c / 1e3, 100.0 * t / tot))
# concurrency of the finest-level launches and of the coarse solves over the wall time
def concurrency(pred):
    ev = []
    for s, e, n, q, g in rows:
        if pred(n, g):
            ev += [(s, 1), (e, -1)]
    ev.sort()
    hist, cur, last = collections.defaultdict(int), 0, rows[0][0]
    for t, d in ev:
        hist[cur] += t - last
        last, cur = t, cur + d
    hist[cur] += rows[-1][1] - last
    return hist
for label, pred in (("finest-level launches in flight", lambda n, g: "strip2d_cross" in n or ("strip2d" in n and g >= 400 * 256)),
                    ("coarse solves in flight", lambda n, g: "cg1_band2d" in n or "cg_reg2d" in n),
                    ("any kernel in flight", lambda n, g: True)):
    h = concurrency(pred)
    print(label + ": " + ", ".join("%d: %.0f %%" % (k, 100.0 * v / span) for k, v in sorted(h.items()) if v > 0.005 * span))
# per queue: busy vs gaps
by_q = collections.defaultdict(list)
for s, e, n, q, g in rows:
    by_q[q].append((s, e))
busy = sum(e - s for v in by_q.values() for s, e in v)
print("per-queue busy %.0f %% of (queues x window)" % (100.0 * busy / (len(by_q) * span)))
