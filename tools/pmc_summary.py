"""Per-kernel averages of rocprofv3 --pmc counter CSVs + FETCH/WRITE calibration on the copy probe.
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]"""
import collections
import csv
import json
import sys

COPY_BYTES = 64 * 1024 * 1024 * 4      # bytes read (= bytes written) by one copy_probe_kernel launch in tools/pmc_run.py


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    out = {"copy_probe_bytes_each_way": COPY_BYTES}
    f_raw = sum(fetch["copy_probe_kernel"]["FETCH_SIZE"]) / len(fetch["copy_probe_kernel"]["FETCH_SIZE"])
    w_raw = sum(write["copy_probe_kernel"]["WRITE_SIZE"]) / len(write["copy_probe_kernel"]["WRITE_SIZE"])
    # counter unit is KiB (x1024); calibration factor = true bytes / (raw * 1024)
    out["fetch_factor"] = COPY_BYTES / (f_raw * 1024.0)
    out["write_factor"] = COPY_BYTES / (w_raw * 1024.0)
    out["kernels"] = {}
    for k in sorted(set(fetch) | set(write)):
        fr = fetch.get(k, {}).get("FETCH_SIZE", [])
        wr = write.get(k, {}).get("WRITE_SIZE", [])
        # skip warm-up launches: use the last half
        fr, wr = fr[len(fr) // 2:], wr[len(wr) // 2:]
        e = {"launches": max(len(fr), len(wr))}
        if fr:
            e["FETCH_SIZE_raw_avg"] = sum(fr) / len(fr)
            e["hbm_read_bytes"] = e["FETCH_SIZE_raw_avg"] * 1024.0 * out["fetch_factor"]
        if wr:
            e["WRITE_SIZE_raw_avg"] = sum(wr) / len(wr)
            e["hbm_write_bytes"] = e["WRITE_SIZE_raw_avg"] * 1024.0 * out["write_factor"]
        out["kernels"][k] = e
    txt = json.dumps(out, indent=1, sort_keys=True)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
