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


def test_recorded_line_has_the_contract_fields():
    """profiles/r2_bench_line.json (the driver-style run of this round) carries every field the contract names."""
    path = os.path.join(ROOT, "profiles", "r2_bench_line.json")
    line = [l for l in open(path) if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"].startswith("BASELINE configs[3]") and d["dtype"] == "f32" and d["n_gpus"] == 1
    assert abs(d["value"] - d["steps"] * d["config"]["per_gpu_batch"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert {e["workload"][:19] for e in d["extra"]} >= {"BASELINE configs[1]", "BASELINE configs[2]"}
