"""Kernel-trace timeline analysis: busy time, idle gaps and per-kernel totals inside a window of a rocprofv3 kernel trace.
usage: python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = ev[int(len(ev) * skip):]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = 0; cur_end = t0; gaps = []
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    if s > cur_end:
        gaps.append(s - cur_end)
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
    k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    agg[k][0] += 1; agg[k][1] += e - s
print("window %.3f ms, busy %.3f ms (%.1f%%), %d kernels, idle gaps: n=%d total %.3f ms mean %.2f us"
      % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), len(ev), len(gaps), sum(gaps) / 1e6, (sum(gaps) / max(len(gaps), 1)) / 1e3))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-62s n=%6d total %8.3f ms avg %7.2f us" % (k, n, t / 1e6, t / n / 1e3))
