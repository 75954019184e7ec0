"""Matrix-pipe utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE counter CSV (one pass, kernel trace
only -- tools/collect_profiles_r4.sh).  MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 4 SIMDs x CUs the counter sums
over); the counter is summed over the shader engines by rocprofv3, so the denominator uses the device's 256 CUs x 4 SIMDs.  Printed per
kernel: launches, average busy cycles, average active cycles, busy / (active x 1024).
usage: python tools/pmc_mfma_busy.py <counter_collection.csv>"""
import collections
import csv
import sys

SIMDS = 256 * 4


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("%-52s %8s %16s %16s %10s" % ("kernel", "launches", "MFMA busy cycles", "GUI active cyc", "busy frac"))
    rows = []
    for k, c in agg.items():
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", []), c.get("GRBM_GUI_ACTIVE", [])
        if not busy or not act:
            continue
        b, a = sum(busy) / len(busy), sum(act) / len(act)
        rows.append((sum(act), k, len(busy), b, a, b / (a * SIMDS) if a else 0.0))
    for _, k, n, b, a, f in sorted(rows, reverse=True)[:24]:
        print("%-52s %8d %16.0f %16.0f %10.3f" % (k[:52], n, b, a, f))
    print("\nbusy frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x %d SIMDs): the share of SIMD-cycles with the matrix pipe busy while the "
          "kernel ran (whole chip in the denominator: a half-chip launch tops out at 0.5)." % SIMDS)


if __name__ == "__main__":
    main()
