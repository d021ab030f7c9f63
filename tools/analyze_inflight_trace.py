"""Developer tool: rocprofv3 --kernel-trace of a candidates-in-flight run -> how long the kernels of one stream take and wait.
usage: python tools/analyze_inflight_trace.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by_q = collections.defaultdict(list)
for r in rows:
    by_q[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
print("queues:", len(by_q))
cg = [e - s for q in by_q.values() for s, e, n in q if "cg_reg2d" in n]
if cg:
    cg.sort()
    print("cg_reg2d: n=%d median %.0f us, p10 %.0f, p90 %.0f" % (len(cg), cg[len(cg)//2]/1e3, cg[len(cg)//10]/1e3, cg[9*len(cg)//10]/1e3))
busy = gap = 0
gaps_after_cg, gaps_before_cg, other_dur = [], [], collections.defaultdict(list)
for q, ev in by_q.items():
    ev.sort()
    for i, (s, e, n) in enumerate(ev):
        busy += e - s
        if "cg_reg2d" not in n:
            other_dur[n[:60]].append(e - s)
        if i:
            g = s - ev[i - 1][1]
            gap += max(g, 0)
            if "cg_reg2d" in ev[i - 1][2]: gaps_after_cg.append(g)
            if "cg_reg2d" in n: gaps_before_cg.append(g)
print("per-queue busy %.1f ms, gaps %.1f ms" % (busy / 1e6, gap / 1e6))
for name, v in (("gap before cg", gaps_before_cg), ("gap after cg", gaps_after_cg)):
    if v:
        v.sort(); print("%s: median %.1f us p90 %.1f us" % (name, v[len(v)//2]/1e3, v[9*len(v)//10]/1e3))
tot = sorted(((sum(v), k, len(v)) for k, v in other_dur.items()), reverse=True)[:8]
for t, k, n in tot:
    print("%-62s n=%6d mean %.1f us total %.1f ms" % (k, n, t / n / 1e3, t / 1e6))
