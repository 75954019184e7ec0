"""Per-kernel averages of every counter CSV under a directory of rocprofv3 --pmc runs (tools/r6_pmc_recur.sh): one line per
(kernel family, counter): launches, average value per launch (warm-up launch skipped).  usage: python tools/pmc_recur_summary.py <dir>"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:64]


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "lstm_persist"
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if flt in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        print("== %s" % k)
        for c, vals in sorted(agg[k].items()):
            v = vals[1:] if len(vals) > 1 else vals
            print("  %-44s launches %2d  avg %18.1f" % (c, len(vals), sum(v) / len(v)))
    if os.path.exists(os.path.join(root, "unprofiled.txt")):
        print("== un-profiled timing")
        print(open(os.path.join(root, "unprofiled.txt")).read())


if __name__ == "__main__":
    main()
