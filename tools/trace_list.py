"""Lists the long kernels of the last third of a rocprofv3 kernel trace in launch order (name, grid, duration).
usage: python tools/trace_list.py <kernel_trace.csv> [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mn = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))) for r in rows)
ev = ev[int(len(ev) * 0.67):]
t0 = ev[0][0]
for s, e, n, g, w in ev:
    if (e - s) / 1e3 >= mn:
        k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:56]
        print("%9.3f ms  %8.1f us  grid %-9s %s" % ((s - t0) / 1e6, (e - s) / 1e3, g, k))
