"""CPU tests of bench.py's contract pieces that do not need a GPU: argument defaults, the self-launch under torch.distributed.run
(VERDICT r1 #1a), the algorithmic work per workload against BASELINE.md section 3, and the JSON schema of a recorded line."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_defaults_name_the_frame_level_headline(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert a.gpus == 1 and a.workload == "lstm" and a.dtype == "f32"
    w = b.WORKLOADS["lstm"]
    assert "configs[3]" in w["name"] and w["batch"] == 128 and w["frame"]
    assert "configs[1]" in b.WORKLOADS["moe"]["name"] and "configs[2]" in b.WORKLOADS["netvlad"]["name"]


def test_algorithmic_work_matches_baseline_md():
    """LSTM 2x1024 x 300 frames = 10 380 902 400 FLOP / video forward; MoE M=2 on 1152 = 54 328 320,
    on 4096 = 193 167 360; NetVLAD 44 236 800 + 44 236 800."""
    b = _bench()
    H, L, F, D, V, M = b.LSTM_H, b.LSTM_L, b.FRAMES, b.D_IN, b.VOCAB, b.MIX
    fwd_lstm = 2.0 * F * ((D + H) * 4 * H + (H + H) * 4 * H)
    assert fwd_lstm == 10380902400.0
    f = b.lstm_flops(1)
    proj_fwd = 2.0 * F * (D * 4 * H + H * 4 * H)
    rec_fwd = 2.0 * F * L * H * 4 * H
    assert proj_fwd + rec_fwd == fwd_lstm
    assert f["lstm_recurrence"] == rec_fwd and f["lstm_recurrence_bwd"] == rec_fwd   # forward / backward recurrent products
    head = 3 * 2.0 * (2 * L * H) * V * (2 * M + 1)
    assert 2.0 * 4096 * V * (2 * M + 1) == 193167360.0
    assert f["gemm"] == proj_fwd + (fwd_lstm) + 2.0 * F * 4 * H * H + head      # projections + dW (= forward FLOPs) + dx1 + head
    assert b.moe_flops(1)["gemm"] == 2 * 54328320.0
    assert b.netvlad_flops(1)["netvlad"] == 2 * (44236800.0 + 44236800.0)


def test_gpus_n_relaunches_itself(monkeypatch):
    """`python bench.py --gpus 4` without a torchrun environment starts 4 ranks through torch.distributed.run with the loopback
    rendezvous and forwards every argument; inside a torchrun environment it does not."""
    b = _bench()
    calls = []
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.setattr(b.subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    with pytest.raises(SystemExit) as e:
        b.maybe_relaunch(b.parse())
    assert e.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert env.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    monkeypatch.setenv("WORLD_SIZE", "4")
    b.maybe_relaunch(b.parse())                                      # already under a launcher: returns
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    b.maybe_relaunch(b.parse())                                      # N = 1: runs in-process
    assert len(calls) == 1


def _latest_line():
    """(name, full record).  Round 5 on: bench.py prints a compact line (profiles/r5_bench_line.json, checked by
    test_round5_line_is_compact_and_names_its_sidecar) and writes the full record to the sidecar it names -- the detailed field
    checks below run on the sidecar."""
    for name in ("r5_bench_extra.json", "r4_bench_line.json", "r3_bench_line.json", "r2_bench_line.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            if name.endswith("_extra.json"):
                return name, json.load(open(path))
            line = [l for l in open(path) if l.startswith("{")][-1]
            return name, json.loads(line)
    raise AssertionError("no recorded bench line under profiles/")


def test_round5_line_is_compact_and_names_its_sidecar():
    """The line the driver-style run of round 5 printed (profiles/r5_bench_line.json): ONE line under 8 KB with every contract field,
    `roofline` and `cpu_baseline` included, equal to what compact_line() makes of the recorded sidecar."""
    path = os.path.join(ROOT, "profiles", "r5_bench_line.json")
    if not os.path.exists(path):
        pytest.skip("no round-5 line recorded yet")
    text = open(path).read().strip()
    assert text.count("\n") == 0 and len(text) < 8192
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "sidecar"):
        assert k in d, k
    assert d["metric"] == "training videos/sec" and d["unit"] == "videos/s" and d["n_gpus"] == 1 and d["dtype"] == "f32"
    assert d["config"]["workload"].startswith("BASELINE configs[3]") and d["config"]["per_gpu_batch"] == 128
    assert abs(d["value"] - d["config"]["per_gpu_batch"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] == "TFLOP/s" and r["traffic"] > 0
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["timed_steps"] >= 10
    g = d["gap_at_20"]
    assert g["frame_twin_lstm"]["within_target"] and g["cpu_twin"]["within_target"] and g["cpu_twin_full_size"]["within_target"]
    full = json.load(open(os.path.join(ROOT, "profiles", "r5_bench_extra.json")))
    assert _bench().compact_line(full) == d


def test_recorded_line_is_small():
    """the driver keeps ~8.9 KB of stdout, so the ONE line bench.py prints must stay under 8 KB whatever the full
    record holds.  The r4 record (24.7 KB, unparsed by the driver) goes through the same trimming function bench.py prints through;
    the result must fit, keep `roofline` and `cpu_baseline` with their contract fields, and name the sidecar with the rest."""
    b = _bench()
    full = json.loads([l for l in open(os.path.join(ROOT, "profiles", "r4_bench_line.json")) if l.startswith("{")][-1])
    assert len(json.dumps(full)) > 20000
    out = b.compact_line(full)
    text = json.dumps(out)
    assert len(text) < 8192 and "\n" not in text
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "sidecar"):
        assert k in out, k
    assert out["value"] == full["value"] and out["ms_per_step"] == full["ms_per_step"]
    r, c = out["roofline"], out["cpu_baseline"]
    assert r["bound"] == "mfma" and r["kernel"] == "lstm_recurrence_bwd" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["unit"] == "TFLOP/s" and "traffic" in r and set(r["families"]) == set(full["roofline"]["families"])
    assert all(set(v) == {"frac", "ms_per_step", "peak"} for v in r["families"].values())
    assert c["kind"] == "port" and c["cores"] == 16 and c["value"] > 0 and c["batch"] == 32 and c["implementation"] == "nn_lstm_twin"
    assert out["gap_at_20"]["cpu_twin_full_size"]["within_target"] and len(out["library"]["sha256"]) == 64
    assert len(out["extra"]) == len(full["extra"]) and out["sidecar"] == b.SIDECAR
    # a record ten times as verbose still fits: the optional detail is shed, the contract fields never are
    fat = dict(full, extra=full["extra"] * 12, library=dict(full["library"], env={"YT8M_%d" % i: "x" * 40 for i in range(60)}))
    small = b.compact_line(fat)
    assert len(json.dumps(small)) < 8192 and small["roofline"]["frac"] == r["frac"] and small["cpu_baseline"]["value"] == c["value"]


def test_recorded_line_has_the_contract_fields():
    """The newest driver-style line under profiles/ carries every field the contract names (round 3 on: also the honesty fields
    of VERDICT r2 #5 -- chip-level fraction next to the occupied-CU one, the blended bound, where `traffic` comes from, the CPU
    baseline's step count and all-cores twin, the identity of the library that produced the numbers)."""
    name, d = _latest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"].startswith("BASELINE configs[3]") and d["dtype"] == "f32" and d["n_gpus"] == 1
    assert abs(d["value"] - d["steps"] * d["config"]["per_gpu_batch"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert {e["workload"][:19] for e in d["extra"]} >= {"BASELINE configs[1]", "BASELINE configs[2]"}
    if name.startswith("r2"):
        return
    assert r["traffic"] is None or "NOT measured in this run" in r["traffic_source"] or "this run" in r["traffic_source"]
    assert r["blended_bound"]["ms_per_step"] > 0 and 0 < r["blended_bound"]["frac"] <= 1.0
    bwd = r["families"]["lstm_recurrence_bwd"]
    assert bwd["occupied_cus"] == 128 and abs(bwd["frac_of_occupied_cus"] - 2 * bwd["frac"]) < 1e-9
    if abs(bwd["peak"] - 2500.0 / 3.0) < 1e-6:                      # round 5: the f16 form (three products) is priced against the f16 pipe ...
        ex = r["exchange"]                                          # ... and says what it really runs against: dz through one CU's L2 port
        assert "16x16x32_f16" in bwd["peak_is"] and 0 < ex["frac"] <= 1.0 and ex["peak_GBps_per_cu"] == 64.0 * 2.1
        assert abs(ex["achieved_GBps_per_cu"] - ex["bytes_per_workgroup_and_step"] / (ex["us_per_step"] * 1e-6) / 1e9) < 1e-6
    else:
        assert bwd["peak"] == 157.3
    if name.startswith("r3"):
        assert c["timed_steps"] >= 10 and (c["all_cores"] is None or c["all_cores"]["cores"] == c["usable_cores"])
    else:                                                           # round 4: see test_round4_line_fields
        _round4_fields(d)
    lib = d["library"]
    assert len(lib["sha256"]) == 64 and lib["in_tree_default"] and isinstance(lib["env"], dict)
    assert "cpu_twin_full_size" in d["gap_at_20"] and d["gap_at_20"]["cpu_twin_full_size"]["within_target"]


def _round4_fields(d):
    """cpu_baseline at B = 32 with the thread count probed and both implementations of the step; NetVLAD's two streaming kernels
    against the HBM roof with the north-star note; the per-GPU batch sweep of the headline as extra lines."""
    c = d["cpu_baseline"]
    assert c["batch"] == 32 and set(c["implementations"]) == {"per_frame_port", "nn_lstm_twin"} and c["implementation"] in c["implementations"]
    assert c["value"] == max(v["value"] for v in c["implementations"].values())
    assert c["implementations"]["nn_lstm_twin"]["timed_steps"] >= 10 and all(v["cores"] >= 1 for v in c["implementations"].values())
    nv = next(e for e in d["extra"] if e["workload"].startswith("BASELINE configs[2]"))
    hbm = nv["roofline"]["hbm"]
    assert set(hbm["kernels"]) == {"vlad_rows", "vlad_cols"} and "41 %" in hbm["north_star_note"] and "82 %" in hbm["north_star_note"]
    for k in hbm["kernels"].values():
        assert 0 < k["frac_of_hbm"] <= 1.0 and abs(k["frac_of_hbm"] - k["achieved_GBps"] / 8000.0) < 1e-12 and k["algorithmic_bytes_per_launch"] > 0
    sweep = {e["per_gpu_batch"]: e for e in d["extra"] if "batch sweep" in e["workload"]}
    assert set(sweep) == {256, 512}
    for e in sweep.values():
        assert e["workload"].startswith("BASELINE configs[3]") and e["dtype"] == "f32" and "lstm_recurrence_bwd" in e["roofline"]["families"]


def test_reducer_trace_report_shape():
    """parallel.GradReducer.trace_report(): nothing traced -> None (the bench line then carries dp_trace: null)."""
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.load_package()
    import yt8m_amd.parallel as parallel
    r = parallel.GradReducer()
    assert r.trace is False and r.trace_report() is None


def test_dp_flags_reach_the_environment(monkeypatch):
    b = _bench()
    for k in ("YT8M_DP_ALGO", "YT8M_DP_RESERVED_CUS", "YT8M_DP_LAYER_BUCKETS"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--dp-algo", "rs_ag", "--dp-reserved-cus", "32", "--dp-layer-buckets", "1", "--dp-bucket-mb", "64"])
    a = b.parse()
    assert os.environ["YT8M_DP_ALGO"] == "rs_ag" and os.environ["YT8M_DP_RESERVED_CUS"] == "32" and os.environ["YT8M_DP_LAYER_BUCKETS"] == "1"
    assert a.dp_bucket_mb == 64
    for k in ("YT8M_DP_ALGO", "YT8M_DP_RESERVED_CUS", "YT8M_DP_LAYER_BUCKETS"):
        monkeypatch.delenv(k, raising=False)                          # parse() wrote os.environ directly


def test_family_peaks_follow_the_pipe_the_kernel_issues_on():
    """x3 GEMMs against 2500 / 6, the one-plane form against 2500 / 3, the fused NetVLAD pooling against the f16 pipe
    (hi + lo: 2500 / 2; single operand in bf16 mode: 2500), the bf16-variant recurrences against 2500, fp32 MFMA kernels against
    157.3 -- and the half-chip backward recurrence against the WHOLE chip with the occupied-CU fraction beside it."""
    b = _bench()
    pk = lambda *a, **k: b.family_peak(*a, **k)[0]
    assert pk("gemm", False) == 157.3 and pk("gemm", True) == 2500.0
    assert abs(pk("gemm_x3", False) - 2500.0 / 6) < 1e-9 and abs(pk("gemm_x1x3", False) - 2500.0 / 3) < 1e-9
    assert abs(pk("gemm_h2", False) - 2500.0 / 3) < 1e-9
    assert pk("netvlad", False) == 1250.0 and pk("netvlad", True) == 2500.0
    assert pk("lstm_recurrence", False) == 157.3 and abs(pk("lstm_recurrence", False, fwd_x3=True) - 2500.0 / 6) < 1e-9
    assert pk("lstm_recurrence", True) == 2500.0 and pk("lstm_recurrence_bwd", True) == 2500.0 and pk("lstm_recurrence_bwd", False) == 157.3
    fam = {"lstm_recurrence_bwd": {"launches_per_step": 6.0, "ms_per_step": 15.0, "avg_launch_ms": 2.5, "declared_flops_per_step": 6.4e11},
           "gemm_x3": {"launches_per_step": 11.0, "ms_per_step": 13.0, "avg_launch_ms": 1.2, "declared_flops_per_step": 2.3e12},
           "optimizer": {"launches_per_step": 2.0, "ms_per_step": 0.7, "avg_launch_ms": 0.35}}
    r = b.roofline_from(fam, {}, False, bwd_cus=128, fwd_x3=True, step_ms=24.0)
    assert r["kernel"] == "lstm_recurrence_bwd" and r["peak"] == 157.3 and r["occupied_cus"] == 128
    assert abs(r["frac"] - 6.4e11 / 15e-3 / 1e12 / 157.3) < 1e-12 and abs(r["frac_of_occupied_cus"] - 2 * r["frac"]) < 1e-12
    bound = 6.4e11 / 157.3e12 * 1e3 + 2.3e12 / (2500e12 / 6) * 1e3
    assert abs(r["blended_bound"]["ms_per_step"] - bound) < 1e-9 and abs(r["blended_bound"]["frac"] - bound / 24.0) < 1e-12
    assert "optimizer" in r["other_families"] and r["traffic"] is None and r["traffic_source"] is None


def test_library_identity_names_the_in_tree_build(monkeypatch):
    b = _bench()
    monkeypatch.setenv("YT8M_SOME_KNOB", "7")
    ident = b.library_identity()
    assert ident["in_tree_default"] and ident["path"].endswith("libyt8m_hip.so") and len(ident["sha256"]) == 64
    assert ident["env"].get("YT8M_SOME_KNOB") == "7"
