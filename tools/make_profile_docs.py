"""Turns the raw files of tools/collect_profiles.sh (gpurun_out/prof_r1/) into the committed summaries under profiles/.
usage: python tools/make_profile_docs.py   (run in the repo root after the gpurun call has merged gpurun_out/)"""
import csv
import json
import os
import shutil
import subprocess

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
O = R + "gpurun_out/prof_r1/"


def sh(c):
    return subprocess.run(c, shell=True, capture_output=True, text=True, cwd=R).stdout


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main():
    shutil.copy(O + "bench/bench_kernel_stats.csv", R + "profiles/r1_bench_kernel_stats.csv")
    bench, bf = last_json(O + "bench_line.json"), last_json(O + "bench_bf16_line.json")
    tab = sh("python tools/prof_summary.py gpurun_out/prof_r1/bench/bench_kernel_trace.csv | head -22")
    rows = {}
    for rec in csv.DictReader(open(O + "bench/bench_kernel_stats.csv")):
        if "gemm_grouped_kernel" in rec["Name"]:
            rows[rec["Name"]] = float(rec["AverageNs"])
    avg = sum(rows.values()) / max(len(rows), 1) / 1e3
    r = bench["roofline"]
    open(R + "profiles/r1_bench_kernel_trace.md", "w").write(
        "# Round 1 -- rocprofv3 --kernel-trace --stats of the bench workload (1 x MI355X)\n\n"
        "Command (round-end build; `tools/collect_profiles.sh` + `tools/make_profile_docs.py`, raw per-kernel stats in "
        "`r1_bench_kernel_stats.csv`):\n\n```\nrocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1/bench -o bench -- "
        "python bench.py --steps 50 --warmup 10 --no-cpu-baseline\n```\n\n"
        "The bench line printed by THIS profiled run: %.0f videos/s, %.3f ms/step, dominant kernel (grouped fp32 GEMM, 2 launches\n"
        "per step) hipEvent average %.4f ms/launch = %.1f TFLOP/s = %.3f of the 157.3 TFLOP/s fp32 matrix peak.\n"
        "rocprofv3's own average over the grouped GEMM launches: %.1f us -> 55.63 GFLOP / %.1f us = %.1f TFLOP/s, i.e. the live\n"
        "hipEvent figure and the profiler agree.\n\n"
        "Per step: 2 grouped GEMM launches (fwd gates+experts; bwd dWg+dWe, the latter on the float4-epilogue instantiation) +\n"
        "split-K fix-up, fused mixing+cross-entropy fwd and bwd, final loss reduction, expert-bias column sum, input L2-normalise,\n"
        "and the 2-pass clip+Adam.\n\n%s\n"
        "bf16-operand VARIANT of the same step (`python bench.py --dtype bf16`, reported separately, never the headline):\n"
        "%.0f videos/s, %.3f ms/step, grouped bf16 GEMM %.1f TFLOP/s (%.3f of the 2.5 PFLOP/s dense bf16 peak) -- the step is then\n"
        "bound by the fp32 cast / mixing / Adam passes (elementwise %.3f ms + optimiser %.3f ms of %.3f ms).\n"
        % (bench["value"], bench["ms_per_step"], r["avg_launch_ms"], r["achieved"], r["frac"], avg, avg, 55632.2 / avg, tab,
           bf["value"], bf["ms_per_step"], bf["roofline"]["achieved"], bf["roofline"]["frac"],
           bf["roofline"]["other_families"]["elementwise"]["ms_per_step"], bf["roofline"]["other_families"]["optimizer"]["ms_per_step"],
           bf["ms_per_step"]))
    gaps = sh("python tools/trace_gaps.py gpurun_out/prof_r1/lstm/lstm_kernel_trace.csv 0.6")
    lstep = [l for l in open(O + "lstm_step.txt").read().splitlines() if "B=" in l][-1]
    open(R + "profiles/r1_lstm_timeline.md", "w").write(
        "# Round 1 -- LstmModel (BASELINE configs[3]) step timeline (1 x MI355X, B = 128, F = 300, 2 x 1024 cells, fp32)\n\n"
        "```\nrocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r1/lstm -o lstm -- python tools/model_bench.py lstm\n"
        "python tools/trace_gaps.py gpurun_out/prof_r1/lstm/lstm_kernel_trace.csv 0.6        # last 40 %% of the trace = ~3 steps\n```\n"
        "Step (profiled run, layer-pipelined stack with 4 time chunks): `%s`\n\n```\n%s```\n"
        "Reading: per training step ~20 ms of hoisted projection / dW / dx GEMMs at the fp32 MFMA roofline (~120 TFLOP/s) and the\n"
        "recurrence: 600 forward step launches (`lstm_step_fwd_kernel<4, 0, 2>`: recurrent product + gates + copy-through) and 600\n"
        "backward step launches (`lstm_step_bwd_kernel<2, 2>`: product of step t + gate backward of step t-1; the pointwise\n"
        "`lstm_gates_bwd_kernel` only opens each time chunk).  The layers of the stack run as a wavefront over time chunks on\n"
        "separate streams (seq_ops._LstmStack), so GEMMs and recurrence overlap (busy fraction above counts overlapped kernels\n"
        "once).  Round history of this step: 54.7 ms (sequential layers) -> 49.5 (layer pipeline) -> 45.5 (float4-of-k packed\n"
        "weights, line-coalesced A tile through LDS, prefetched epilogue operands) -> 42.2-42.6 (one launch per backward step).\n"
        "What bounds the step kernels now (DESIGN.md section 4, 'Recurrence step kernels'): an L2-bandwidth-bound K loop (128 MB of\n"
        "h / W_h per step) plus 6-7 us of fixed cost per launch, against a 6.8 us MFMA bound.\n" % (lstep, gaps))
    mb = "\n".join(l for l in open(O + "model_bench.txt").read().splitlines() if "B=" in l)
    open(R + "profiles/r1_plugin_step_times.md", "w").write(
        "# Round 1 -- training-step time of every plugin configuration (1 x MI355X, synthetic inputs, un-profiled by rocprofv3)\n\n"
        "`python tools/model_bench.py` (+ `config5_bf16 netvlad_bf16`): 5 timed steps after 2 warm-up steps; frame-level models get raw\n"
        "uint8 [B,300,1152] input, video-level models fp32 [B,1152]; V = 4716; families are hipEvent sums inside the library (they\n"
        "overlap for the layer-pipelined LSTM stacks, so their sum exceeds the step time there; the library profiler also turns the\n"
        "hipGraph replay off: see the `lstm(no library profiler ...)` row).\n\n```\n%s\n```\n"
        "Default `python bench.py` line of the same build: %.1f k videos/s, %.3f ms/step, roofline.frac %.3f.\n\n"
        "fp32 grouped GEMM on the shapes of these models (`tools/gemm_shapes.py`):\n\n```\n%s\n```\n"
        "Streaming kernels of the attention-logit layer (`tools/skinny_bench.py`; TB/s = 4*M*K bytes / time):\n\n```\n%s\n```\n"
        % (mb, bench["value"] / 1e3, bench["ms_per_step"], r["frac"],
           "\n".join(l for l in open(O + "gemm_shapes.txt").read().splitlines() if "TFLOP" in l),
           "\n".join(l for l in open(O + "skinny_bench.txt").read().splitlines() if "M=" in l)))
    print("profiles/ refreshed")


if __name__ == "__main__":
    main()
