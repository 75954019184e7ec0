"""Per-kernel averages of ONE rocprofv3 --pmc counter CSV, for the kernels whose name contains a filter string.
usage: python tools/pmc_kernels.py <counter_collection.csv> <name filter>
FETCH_SIZE / WRITE_SIZE are in KiB on this rocprofv3; MI355X_MICROARCH.md (HBM): FETCH_SIZE reports half of a wide coalesced
streaming read on gfx950 -- the x2 correction is applied by whoever quotes the number (this tool prints raw and x2)."""
import collections
import csv
import sys


def main():
    path, flt = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if flt in name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name in sorted(agg):
        for c, vals in sorted(agg[name].items()):
            v = vals[len(vals) // 2:]                      # skip the warm-up launches
            avg = sum(v) / len(v)
            print("%-44s %-12s launches %3d  avg raw %12.1f KiB = %8.1f MB  (x2: %8.1f MB)" % (name[:44], c, len(vals), avg,
                                                                                              avg * 1024 / 1e6, 2 * avg * 1024 / 1e6))


if __name__ == "__main__":
    main()
