"""Builds the round-2 profile summaries under profiles/ from gpurun_out/prof_r2/ (tools/collect_profiles_r2.sh)."""
import contextlib
import csv
import io
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "prof_r2")
P = os.path.join(R, "profiles")
sys.path.insert(0, os.path.join(R, "tools"))


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def stats_table(path, steps, top=16):
    rows = list(csv.DictReader(open(path)))
    out = ["| kernel | launches / step | avg us | total ms / step | % of GPU kernel time |", "|---|---|---|---|---|"]
    for r in rows[:top]:
        out.append("| `%s` | %.1f | %.1f | %.3f | %s |" % (short(r["Name"])[:80], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                          float(r["TotalDurationNs"]) / 1e6 / steps, r["Percentage"]))
    return "\n".join(out)


def last_json(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def main():
    shutil.copy(os.path.join(O, "bench", "bench_kernel_stats.csv"), os.path.join(P, "r2_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(O, "moe", "moe_kernel_stats.csv"), os.path.join(P, "r2_moe_kernel_stats.csv"))
    shutil.copy(os.path.join(O, "netvlad", "nv_kernel_stats.csv"), os.path.join(P, "r2_netvlad_kernel_stats.csv"))
    for f, t in (("bench_line.json", "r2_bench_line.json"), ("persist_check.txt", "r2_persist_check.txt"),
                 ("gemm_shapes_lstm.txt", "r2_gemm_shapes_lstm.txt"), ("model_bench.txt", "r2_plugin_step_times.txt"),
                 ("x3_check.txt", "r2_x3_check.txt"), ("probe_mfma.txt", "r2_probe_mfma.txt")):
        shutil.copy(os.path.join(O, f), os.path.join(P, t))
    import pmc_summary
    sys.argv = ["pmc_summary", os.path.join(O, "pmc", "fetch_counter_collection.csv"), os.path.join(O, "pmc", "write_counter_collection.csv")]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        pmc_summary.main()
    pm = json.loads(buf.getvalue())
    B, F, D, H = 128, 300, 1152, 1024
    k = pm["kernels"]

    def tot(name):
        e = k[name]
        return e.get("hbm_read_bytes", 0.0) + e.get("hbm_write_bytes", 0.0)

    T = F // 2                                     # steps per chunk at lstm_pipeline_chunks = 2
    fam = {}

    def pick(prefix):
        ks = [n for n in k if n.startswith(prefix)]
        return max(ks, key=lambda n: k[n]["launches"]) if ks else None

    # x3 GEMMs of the headline step (per step: 2 projections of layer 1, 2 dx, 2 + 2 grouped weight-gradient launches, 1 head
    # product); operands are x3 images (6 B / element), C is fp32
    def x3_bytes(M, N, K):
        return 6.0 * (M * K + N * K) + 4.0 * M * N

    shapes = [(T * B, 4 * H, H)] * 2 + [(T * B, H, 4 * H)] * 2 + [(H, 4 * H, T * B), (H, 4 * H, T * B)] * 2 + \
             [(D, 4 * H, T * B), (H, 4 * H, T * B)] * 2
    launches = 2 + 2 + 2 + 2
    g = pick("gemm_x3_kernel<3>")
    if g:
        fam["gemm_x3"] = {"kernel": g, "hbm_bytes_per_launch": tot(g),
                          "algorithmic_bytes_per_launch": sum(x3_bytes(*sh) for sh in shapes) / launches,
                          "note": "average over the step's x3 launches (projection, dx and grouped weight-gradient products)"}
    f = pick("lstm_persist_fwd")
    algf = float(T * (B * 4 * H * 4 * 2 + 3 * B * H * 4))
    note = ("algorithmic = the saved activations only (z in, gates / c / h / out written); the state exchange (one 768 KB image of "
            "bf16 planes per step forward, 2 MB fp32 backward) is written through once and fetched once per XCD into its L2: these "
            "bytes pass the memory-side counters too (the Infinity Cache serves most of them)")
    # the state exchange crosses the L2s by design: every image is written through once and fetched once by each of the 8 XCDs
    img_f = B * H * (6 if f and "x3" in f else 4)
    img_b = B * 4 * H * 4
    if f:
        fam["lstm_recurrence"] = {"kernel": f, "hbm_bytes_per_launch": tot(f), "algorithmic_bytes_per_launch": algf,
                                  "state_exchange_bytes_per_launch": float(T * img_f * 9), "note": note}
    bk = pick("lstm_persist_bwd_kernel")
    if bk:
        fam["lstm_recurrence_bwd"] = {"kernel": bk, "hbm_bytes_per_launch": tot(bk), "algorithmic_bytes_per_launch": algf,
                                      "state_exchange_bytes_per_launch": float(T * img_b * 9)}
    g32 = pick("gemm_grouped_kernel")
    if g32:
        fam["gemm"] = {"kernel": g32, "hbm_bytes_per_launch": tot(g32), "algorithmic_bytes_per_launch": None,
                       "note": "the MoE head products at B = 128 (fp32 MFMA kernel)"}
    out = {"unit": "bytes/launch (memory-side; PMC FETCH_SIZE x %.3f + WRITE_SIZE x %.3f, separate passes, calibrated on the 256 MiB "
                   "copy probe of the same run as MI355X_MICROARCH.md prescribes)" % (pm["fetch_factor"], pm["write_factor"]),
           "fetch_factor": pm["fetch_factor"], "write_factor": pm["write_factor"], "families": fam, "kernels": pm["kernels"]}
    json.dump(out, open(os.path.join(P, "r2_pmc_traffic_lstm.json"), "w"), indent=1, sort_keys=True)
    line = last_json(os.path.join(O, "bench_line.json"))
    linep = last_json(os.path.join(O, "bench_line_profiled.json"))
    steps = 28.0          # 20 timed + 3 warm-up + 5 hipEvent-profile steps in the traced run
    md = ["# Round 2: headline bench (BASELINE configs[3], LstmModel B=128, fp32) under rocprofv3", "",
          "Command (tools/collect_profiles_r2.sh): `rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 "
          "--no-cpu-baseline --no-gap --no-extra`; the un-profiled driver-style line is `profiles/r2_bench_line.json`.", "",
          "* un-profiled: **%.2f ms/step, %.0f videos/s**; under the tracer: %.2f ms/step." % (line["ms_per_step"], line["value"], linep["ms_per_step"])]
    sl = (line.get("roofline") or {}).get("step_level")
    if sl:
        md.append("* whole-step matrix utilisation: %.1f TFLOP/s = %.2f of the fp32 MFMA peak (all algorithmic FLOPs of the step over "
                  "wall time)." % (sl["achieved"], sl["frac"]))
    md += ["* Launches of different streams share the chip (two half-chip backward recurrences run side by side, the weight-gradient "
           "GEMMs take the CUs they leave), so the per-kernel durations below overlap and include time spent waiting for CUs: they add "
           "up to more than the step.  Stand-alone rates: `profiles/r2_x3_check.txt` (GEMMs), `profiles/r2_persist_check.txt` "
           "(recurrences), `profiles/r2_probe_mfma.txt` (matrix-pipe ceilings of this box).", "",
           stats_table(os.path.join(O, "bench", "bench_kernel_stats.csv"), steps), "",
           "Full table: `profiles/r2_bench_kernel_stats.csv`.  PMC traffic: `profiles/r2_pmc_traffic_lstm.json`.", "",
           "## Extra lines under the tracer", "",
           "configs[1] (`bench.py --workload moe --steps 200`): %.3f ms/step; per-kernel: `profiles/r2_moe_kernel_stats.csv`" % last_json(os.path.join(O, "moe_line.json"))["ms_per_step"],
           "", stats_table(os.path.join(O, "moe", "moe_kernel_stats.csv"), 230.0, 8), "",
           "configs[2] (`bench.py --workload netvlad --steps 20`, B = 1024): %.3f ms/step; per-kernel: `profiles/r2_netvlad_kernel_stats.csv`" % last_json(os.path.join(O, "netvlad_line.json"))["ms_per_step"],
           "", stats_table(os.path.join(O, "netvlad", "nv_kernel_stats.csv"), 28.0, 12), ""]
    open(os.path.join(P, "r2_bench_kernel_trace.md"), "w").write("\n".join(md))
    print("\n".join(md))


if __name__ == "__main__":
    main()
