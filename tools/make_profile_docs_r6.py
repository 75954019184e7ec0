"""Builds the round-6 profile summaries under profiles/ from gpurun_out/prof_r6/ (tools/collect_profiles_r6.sh) and the driver-style
bench line (gpurun_out/r6_full.json, or prof_r6/bench_line.json when that run completed)."""
import contextlib
import csv
import io
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "prof_r6")
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


def trim_stats(src, dst, top=40):
    """kernel_stats.csv with the (very long) templated torch kernel names cut to 160 characters"""
    rows = list(csv.reader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        for r in rows[:top + 1]:
            r[0] = r[0][:160]
            w.writerow(r)


def pmc_traffic():
    """profiles/r6_pmc_traffic_lstm.json from the FETCH_SIZE / WRITE_SIZE passes (run on the GPU box BEFORE the bench line of the same
    collection, which cites it: `python tools/make_profile_docs_r5.py pmc`)."""
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

    def pick(prefix):
        ks = [n for n in k if n.startswith(prefix)]
        return max(ks, key=lambda n: k[n]["launches"]) if ks else None

    fam = {}

    def x3_bytes(M, N, K, pa=3):
        return 2.0 * pa * M * K + 6.0 * N * K + 4.0 * M * N

    # the step's three-product f16 launches (native stack: layer-1 projection over the whole sequence; per backward part of 100 / 100 / 50 /
    # 50 steps: dx, the grouped layer-1 weight gradient (two products), layer 0's h-part weight gradient; the head's weight gradient):
    # h2 images are 4 B per operand element (two half planes), C is fp32
    parts = [100, 100, 50, 50]

    def h2_bytes(M, N, K, pa=2):
        return 2.0 * pa * M * K + 4.0 * N * K + 4.0 * M * N

    shapes = [(F * B, 4 * H, H)] + [(T * B, H, 4 * H) for T in parts] + [(H, 4 * H, T * B) for T in parts for _ in range(3)]
    launches = 1 + 4 + 4 + 4
    g = pick("gemm_h2q_kernel<2>")
    if g:
        fam["gemm_h2"] = {"kernel": g, "hbm_bytes_per_launch": tot(g), "algorithmic_bytes_per_launch": sum(h2_bytes(*sh) for sh in shapes) / launches,
                          "note": "average over the recurrent stack's h2 launches (layer-1 projection, dx and weight-gradient products of the four "
                                  "backward parts; the head's weight-gradient launch of the same kernel is in the measured average, not in this figure)"}
    g1 = pick("gemm_h2q_kernel<1>")
    if g1:
        sh1 = [(F * B, 4 * H, D)] + [(D, 4 * H, T * B) for T in parts]
        fam["gemm_h1x2"] = {"kernel": g1, "hbm_bytes_per_launch": tot(g1), "algorithmic_bytes_per_launch": sum(h2_bytes(*sh, pa=1) for sh in sh1) / 5.0,
                            "note": "one exact half plane of the uint8 frames (2 B / element) against a two-plane operand: layer-0 projection and weight gradient"}
    f = pick("lstm_persist_fwd")
    per_step = float(B * 4 * H * 4 * 2 + 3 * B * H * 4)          # z in, gates / c / h / out written
    note = ("algorithmic = the saved activations only; the state exchange (one image per step: 768 KB of bf16 planes forward, 2 MB of "
            "half planes + scale words backward) is written through once and fetched once per XCD into its L2 -- those bytes pass the memory-side counters too")
    img_f = B * H * (6 if f and "x3" in f else 4)
    img_b = B * 4 * H * 4
    if f:
        fam["lstm_recurrence"] = {"kernel": f, "hbm_bytes_per_launch": tot(f), "algorithmic_bytes_per_launch": F * per_step,
                                  "state_exchange_bytes_per_launch": float(F * img_f * 9), "note": note}
    bk = pick("lstm_persist_bwd_kernel")
    if bk:
        fam["lstm_recurrence_bwd"] = {"kernel": bk, "hbm_bytes_per_launch": tot(bk), "algorithmic_bytes_per_launch": 75 * per_step,
                                      "state_exchange_bytes_per_launch": float(75 * img_b * 9),
                                      "note": "eight launches per step of 100 / 100 / 50 / 50 time steps (two layers): averages per launch = 75 steps"}
    g32 = pick("gemm_grouped_kernel")
    if g32:
        fam["gemm"] = {"kernel": g32, "hbm_bytes_per_launch": tot(g32), "algorithmic_bytes_per_launch": None,
                       "note": "the MoE head products at B = 128 (fp32 MFMA kernel)"}
    out = {"unit": "bytes/launch (memory-side; PMC FETCH_SIZE x %.3f + WRITE_SIZE x %.3f, separate rocprofv3 --pmc passes, calibrated on the "
                   "256 MiB copy probe of the same run as MI355X_MICROARCH.md prescribes)" % (pm["fetch_factor"], pm["write_factor"]),
           "fetch_factor": pm["fetch_factor"], "write_factor": pm["write_factor"], "families": fam, "kernels": pm["kernels"]}
    json.dump(out, open(os.path.join(P, "r6_pmc_traffic_lstm.json"), "w"), indent=1, sort_keys=True)
    return out


def main():
    trim_stats(os.path.join(O, "bench", "bench_kernel_stats.csv"), os.path.join(P, "r6_bench_kernel_stats.csv"))
    trim_stats(os.path.join(O, "moe", "moe_kernel_stats.csv"), os.path.join(P, "r6_moe_kernel_stats.csv"))
    trim_stats(os.path.join(O, "netvlad", "nv_kernel_stats.csv"), os.path.join(P, "r6_netvlad_kernel_stats.csv"))
    trim_stats(os.path.join(O, "c5", "c5_kernel_stats.csv"), os.path.join(P, "r6_config5_bf16_kernel_stats.csv"))
    for f, t in (("persist_check.txt", "r6_persist_check.txt"), ("gemm_shapes_lstm.txt", "r6_gemm_shapes_lstm.txt"),
                 ("model_bench.txt", "r6_plugin_step_times.txt"), ("x3_check.txt", "r6_x3_check.txt"), ("b1_bench.txt", "r6_b1_bench.txt"),
                 ("step_timeline.txt", "r6_step_timeline.txt"), ("reader_bench.txt", "r6_reader_bench.txt"),
                 ("mfma_busy.txt", "r6_pmc_mfma_busy.txt"), ("persist_timeline.txt", "r6_persist_bwd_timeline.txt"),
                 ("fwd_pair_check.txt", "r6_fwd_pair_check.txt")):
        if os.path.exists(os.path.join(O, f)):
            shutil.copy(os.path.join(O, f), os.path.join(P, t))
    line_path = os.path.join(O, "bench_line.json")
    extra_path = os.path.join(O, "bench_extra.json")
    if not os.path.exists(line_path) or os.path.getsize(line_path) == 0:
        # the collection's own default run was the one that exposed the hanging all-cores CPU leg (DESIGN_LOG 11.9): the line is the
        # default `python bench.py` of the fixed build (tools/r6_bench_default.sh, a later gpurun call of the same round)
        line_path = os.path.join(R, "gpurun_out", "r6_bench_default.json")
        extra_path = os.path.join(R, "gpurun_out", "r6_bench_extra.json")
    shutil.copy(line_path, os.path.join(P, "r6_bench_line.json"))            # the ONE line (<= 8 KB) ...
    shutil.copy(extra_path, os.path.join(P, "r6_bench_extra.json"))          # ... and the sidecar it names
    for f, t in (("netvlad_ab.txt", "r6_netvlad_single_pass_ab.txt"), ("netvlad_single_sections.txt", "r6_netvlad_single_pass_sections.txt")):
        if os.path.exists(os.path.join(O, f)):
            shutil.copy(os.path.join(O, f), os.path.join(P, t))
    pmc_traffic()
    line = last_json(os.path.join(P, "r6_bench_line.json"))
    linep = last_json(os.path.join(O, "bench_line_profiled.json"))
    steps = 28.0          # 20 timed + 3 warm-up + 5 hipEvent-profile steps in the traced run
    r = line["roofline"]
    md = ["# Round 6: headline bench (BASELINE configs[3], LstmModel B=128, fp32) under rocprofv3", "",
          "Commands (tools/collect_profiles_r6.sh): the driver-style line `python bench.py` -> `profiles/r6_bench_line.json`; "
          "`rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra` -> the table below.", "",
          "* un-profiled: **%.2f ms/step, %.0f videos/s** (round 5: 16.39 / 7809); under the tracer: %.2f ms/step." % (line["ms_per_step"], line["value"], linep["ms_per_step"]),
          "* dominant kernel family by hipEvent time: `%s`: %.1f TFLOP/s = **%.3f of the whole chip's %s peak** (%.3f of the %s CUs it occupies)"
          % (r["kernel"], r["achieved"], r["frac"], r.get("peak_is", "?").split(":")[0], r.get("frac_of_occupied_cus", float("nan")), r.get("occupied_cus", "?")),
          "* blended matrix bound of the step (sum of family FLOPs / chip peak of the pipe each issues on): %.2f ms = %.2f of the measured step."
          % (r["blended_bound"]["ms_per_step"], r["blended_bound"]["frac"]),
          "* Launches of different streams share the chip (two half-chip backward recurrences run side by side, the weight-gradient GEMMs "
          "take the CUs they leave), so the per-kernel durations below overlap and include time spent waiting for CUs: they add up to more "
          "than the step.  One step as a timeline: `profiles/r6_step_timeline.txt`.  Stand-alone rates: `profiles/r6_x3_check.txt`, "
          "`r6_gemm_shapes_lstm.txt` (GEMMs), `r6_persist_check.txt` (recurrences), `r6_b1_bench.txt` (bf16 image kernel).", "",
          stats_table(os.path.join(O, "bench", "bench_kernel_stats.csv"), steps), "",
          "Full table: `profiles/r6_bench_kernel_stats.csv`.  PMC traffic: `profiles/r6_pmc_traffic_lstm.json`.", "",
          "## Extra lines under the tracer", "",
          "configs[1] (`bench.py --workload moe --steps 200`): %.3f ms/step; per-kernel: `profiles/r6_moe_kernel_stats.csv`" % last_json(os.path.join(O, "moe_line.json"))["ms_per_step"],
          "", stats_table(os.path.join(O, "moe", "moe_kernel_stats.csv"), 230.0, 8), "",
          "configs[2] (`bench.py --workload netvlad --steps 20`, B = 1024): %.3f ms/step; per-kernel: `profiles/r6_netvlad_kernel_stats.csv`" % last_json(os.path.join(O, "netvlad_line.json"))["ms_per_step"],
          "", stats_table(os.path.join(O, "netvlad", "nv_kernel_stats.csv"), 28.0, 12), "",
          "configs[4] in bf16 (`tools/model_bench.py config5_bf16_b1024`, B = 1024, 7 steps traced): " + open(os.path.join(O, "config5_bf16.txt")).read().strip().splitlines()[-1][:90],
          "", stats_table(os.path.join(O, "c5", "c5_kernel_stats.csv"), 7.0, 12), ""]
    open(os.path.join(P, "r6_bench_kernel_trace.md"), "w").write("\n".join(md))
    print("\n".join(md[:12]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pmc":
        os.makedirs(P, exist_ok=True)
        pmc_traffic()
    else:
        main()
