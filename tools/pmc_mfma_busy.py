"""Matrix-pipe utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE counter CSV (one pass, kernel trace
only -- tools/collect_profiles_r4.sh).  Units on gfx950 / rocprofv3 (checked on this box: the backward recurrence's busy count equals
its MFMA instruction count x 32 cycles exactly -- 75 steps x 128 workgroups x 8 waves x 4 items x 128 v_mfma_f32_16x16x4_f32 --
and the 256 MiB copy probe's active count is 8 x its duration in clocks): SQ_VALU_MFMA_BUSY_CYCLES is the sum over all waves of the
cycles their MFMA instructions keep a matrix pipe busy; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  MFMA-busy fraction of the WHOLE
chip while the kernel ran = busy / ((active / 8) x 1024 SIMDs).  Printed per kernel: launches, average busy cycles, average active
cycles (sum over XCDs), the fraction.
usage: python tools/pmc_mfma_busy.py <counter_collection.csv>"""
import collections
import csv
import sys

SIMDS = 256 * 4
XCDS = 8


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
        rows.append((sum(act), k, len(busy), b, a, b / (a / XCDS * SIMDS) if a else 0.0))
    for _, k, n, b, a, f in sorted(rows, reverse=True)[:24]:
        print("%-52s %8d %16.0f %16.0f %10.3f" % (k[:52], n, b, a, f))
    print("\nbusy frac = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / %d XCDs) x %d SIMDs): the share of the chip's SIMD-cycles with the matrix "
          "pipe busy while the kernel ran (whole chip in the denominator: a half-chip launch tops out at 0.5).  Counter passes serialise "
          "the kernels: these are stand-alone figures, not the overlapped step." % (XCDS, SIMDS))


if __name__ == "__main__":
    main()
