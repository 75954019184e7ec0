"""Summarises a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) into a per-kernel table (markdown).
usage: python tools/prof_summary.py <kernel_trace.csv> [--skip-first N]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    for r in rows:
        name = r.get("Kernel_Name") or r.get("kernel_name")
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        key = (name, grid)
        agg.setdefault(key, []).append(dur)
    tot = sum(sum(v) for v in agg.values())
    print("| kernel | grid.x | calls | avg us | min us | max us | % of GPU time |")
    print("|---|---|---|---|---|---|---|")
    for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short[:90]
        print("| `%s` | %s | %d | %.1f | %.1f | %.1f | %.1f |" % (short, grid, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3,
                                                                  max(v) / 1e3, 100.0 * sum(v) / tot))
    print("\ntotal GPU kernel time: %.3f ms over %d dispatches" % (tot / 1e6, len(rows)))


if __name__ == "__main__":
    main()
