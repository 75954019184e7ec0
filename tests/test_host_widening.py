"""CPU tests of the section-8f widening pieces that are pure host logic: CSV line format and checkpoint round trip."""
import numpy as np
import pytest
import torch

import yt8m_amd.checkpoint as checkpoint
import yt8m_amd.inference as inference
import yt8m_amd.train as train
from yt8m_amd.variables import Graph, random_normal, zeros


def test_format_lines_host_path():
    p = np.array([[0.1, 0.9, 0.5, 0.7], [0.25, 0.125, 0.75, 0.5]], dtype=np.float32)
    lines = list(inference.format_lines([b"abc", "xyz"], p, 2))
    assert lines == ["abc,1 0.900000 3 0.700000\n", "xyz,2 0.750000 3 0.500000\n"]
    assert inference.CSV_HEADER == "VideoId,LabelConfidencePairs\n"


def test_checkpoint_round_trip_and_rotation(tmp_path):
    class TG(object):
        pass

    def make(seed):
        g = Graph(device="cpu", seed=seed)
        g.begin_step()
        g.get_variable("gates/weights", (6, 9), random_normal(0.3), l2=1e-8)
        g.get_variable("experts/biases", (9,), zeros)
        g.get_variable("bn/moving_mean", (9,), zeros, trainable=False)
        g.finalize()
        tg = TG()
        tg.graph, tg.global_step = g, 0
        return tg
    a = make(1)
    a.graph.adam_m.normal_()
    a.graph.adam_v.uniform_()
    for step in (10, 20, 30, 40):
        a.global_step = step
        path = checkpoint.save(a, str(tmp_path), max_to_keep=3)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["model.ckpt-20.safetensors", "model.ckpt-30.safetensors",
                                                         "model.ckpt-40.safetensors"]
    assert checkpoint.latest_checkpoint(str(tmp_path)) == path
    b = make(2)
    assert not torch.equal(b.graph.vars["gates/weights"].data, a.graph.vars["gates/weights"].data)
    checkpoint.restore(b, path)
    assert b.global_step == 40
    for k in a.graph.vars:
        assert torch.equal(b.graph.vars[k].data, a.graph.vars[k].data)
    assert torch.equal(b.graph.adam_m[:54], a.graph.adam_m[:54]) and torch.equal(b.graph.adam_v[:54], a.graph.adam_v[:54])
    # documented resume flow: variables exist (one forward pass) but the graph is NOT finalized yet -> restore() must
    # finalize through TrainGraph.ensure_finalized (l2 scaling + reducer attach happen there) and still load the Adam slots
    calls = []
    c = TG()
    c.graph, c.global_step = Graph(device="cpu", seed=3), 0
    c.graph.begin_step()
    c.graph.get_variable("gates/weights", (6, 9), random_normal(0.3), l2=1e-8)
    c.graph.get_variable("experts/biases", (9,), zeros)
    c.graph.get_variable("bn/moving_mean", (9,), zeros, trainable=False)
    c.ensure_finalized = lambda: (calls.append(1), train.TrainGraph.ensure_finalized(c))[1]
    c.reg_penalty, c.reducer = 1, None
    assert not c.graph.finalized
    checkpoint.restore(c, path)
    assert calls == [1] and c.graph.finalized and c.global_step == 40
    assert torch.equal(c.graph.adam_m[:54], a.graph.adam_m[:54]) and torch.equal(c.graph.adam_v[:54], a.graph.adam_v[:54])
    from safetensors.torch import load_file
    sd = load_file(path)
    assert {"gates/weights", "gates/weights/Adam", "gates/weights/Adam_1", "experts/biases", "global_step"} <= set(sd)
    assert tuple(sd["gates/weights"].shape) == (6, 9)


def test_lstm_time_partitions(monkeypatch):
    """Host logic of the LSTM stack's time partition (seq_ops._chunks / _bwd_parts): chunks cover [0, F) exactly once in order;
    an explicit backward partition (fractions or a count) wins; the persistent path's own backward parts apply only when asked
    for, and every variant degrades gracefully when F is smaller than the number of parts."""
    import yt8m_amd.seq_ops as seq_ops

    def covers(parts, F):
        t = 0
        for t0, T in parts:
            assert t0 == t and T > 0
            t += T
        assert t == F

    for F in (1, 2, 7, 24, 300, 301):
        for n in (1, 2, 3, 4, 10, 400):
            parts = seq_ops._chunks(F, n)
            covers(parts, F)
            assert len(parts) <= min(n, F)
    fwd = seq_ops._chunks(300, 2)
    monkeypatch.setattr(seq_ops, "BWD_PARTS", [])
    monkeypatch.setattr(seq_ops, "BWD_CHUNKS", 0)
    monkeypatch.setattr(seq_ops, "PERSIST_BWD_CHUNKS", 3)
    assert seq_ops._bwd_parts(300, fwd, False) == fwd                      # per-step kernels: the caller's partition
    assert seq_ops._bwd_parts(300, fwd, True) == [(0, 100), (100, 100), (200, 100)]
    covers(seq_ops._bwd_parts(2, seq_ops._chunks(2, 2), True), 2)
    monkeypatch.setattr(seq_ops, "PERSIST_BWD_CHUNKS", 0)
    assert seq_ops._bwd_parts(300, fwd, True) == fwd
    monkeypatch.setattr(seq_ops, "BWD_CHUNKS", 4)
    assert seq_ops._bwd_parts(300, fwd, True) == seq_ops._chunks(300, 4)
    monkeypatch.setattr(seq_ops, "BWD_PARTS", [1.0, 2.0, 3.0])
    parts = seq_ops._bwd_parts(300, fwd, True)
    assert parts == [(0, 50), (50, 100), (150, 150)]
    covers(seq_ops._bwd_parts(2, fwd, False), 2)                           # empty parts are dropped


def test_support_label_types(tmp_path, flags):
    """MultiTaskLoss.get_support (W/losses.py:221-257): the vertical table loader (two-integer lines only) equals the oracle's; the
    "label" / "frequent" supports -- pure slicing, host tensors allowed -- equal the oracle; an unknown type raises as the
    reference does."""
    import numpy as np
    import pytest
    import torch
    import yt8m_amd.losses as losses
    from oracle import np_ref
    text = "0 1\n3 2\n\n1 2 3\n5\n3 0\n9 4\n"
    path = tmp_path / "vertical.tsv"
    path.write_text(text)
    vm = losses.load_vertical_mapping(str(path), 10, 5)
    want = np_ref.load_vertical_mapping(text.splitlines(), 10, 5)
    assert vm.dtype == np.float32 and np.array_equal(vm, want) and vm.sum() == 4 and vm[3, 0] == vm[3, 2] == vm[9, 4] == 1
    rs = np.random.RandomState(0)
    y = rs.rand(6, 10) < 0.3
    flags.num_frequents = 4
    got = losses.MultiTaskLoss().get_support(torch.from_numpy(y), "label,frequent,frequent")
    assert got.dtype == torch.float32
    assert np.array_equal(got.numpy(), np_ref.get_support_label_type(y, "label,frequent,frequent", num_frequents=4))
    assert np.array_equal(np_ref.get_support_label_type(y, "vertical", vertical_mapping=want),
                          (y.astype(np.float64) @ want > 0.2).astype(np.float64))
    with pytest.raises(NotImplementedError):
        losses.MultiTaskLoss().get_support(torch.from_numpy(y), "nonsense")


def test_gemm_dispatch_cost_model():
    """ops._x3_wins (host decision between the six-product bf16-pipe GEMM and the fp32-MFMA kernel): the large products of the
    BASELINE configurations take the bf16 pipe, tiny / skinny ones do not, empty problems are ignored, and the decision is made
    per product -- so it cannot depend on how products are grouped into launches (the data-parallel path launches the
    weight-gradient products one by one, the plain step grouped; tests/test_gpu_x3.py checks the bits)."""
    import yt8m_amd.ops as ops
    big = [(19200, 4096, 1152),       # LSTM layer-0 projection of one time chunk
           (38400, 4096, 1024),       # layer-1 projection, whole sequence
           (1152, 4096, 12800),       # weight gradient of one backward part
           (1024, 23580, 1152),       # MoE head at B = 1024
           (1152, 9432, 1024)]        # its gate weight gradient
    small = [(128, 64, 64), (8, 4716, 1024), (300, 77, 50), (128, 23580, 16)]
    for shp in big:
        assert ops._x3_wins([shp], False, False), shp
    for shp in small:
        assert not ops._x3_wins([shp], False, False), shp
    assert not ops._x3_wins([(0, 4096, 1024)], False, False) and not ops._x3_wins([], False, False)
    # few tiles, very long reduction (round 3: up to 16 K parts): the NetVLAD hidden FC goes to the bf16 pipe, the einsum-CNN
    # weight gradients (5 / 9 / 20 tiles at K = 38 400) and the MoE-chain dx shapes (18 tiles at K ~ 14 000) stay where they measured
    # faster (tools/gemm_auto_probe.py, profiles/r3_plugin_step_times.txt)
    assert ops._x3_wins([(1024, 1024, 73728)], False, False)
    for shp in [(1152, 128, 38400), (2304, 128, 38400), (1152, 1024, 38400), (512, 2304, 14148)]:
        assert not ops._x3_wins([shp], False, False), shp
    # monotone in every dimension once it wins
    assert ops._x3_wins([(2 * 19200, 4096, 1152)], False, False) and ops._x3_wins([(19200, 2 * 4096, 1152)], False, False)


def test_early_optimizer_ranges_and_the_complement_the_step_covers():
    """seq_ops._ready_ranges (which variables the recurrent stack's backward pass may update early) and the complement loop of
    train.TrainGraph.step: together they cover every trainable variable exactly once."""
    import yt8m_amd.seq_ops as seq_ops
    for ready in ([False, False, True, True, True], [True, False, True, False, True, True], [True] * 4, [False] * 3, []):
        rng = seq_ops._ready_ranges(ready)
        assert [i for lo, hi in rng for i in range(lo, hi)] == [i for i, r in enumerate(ready) if r]
        assert all(hi > lo for lo, hi in rng) and all(rng[k][1] < rng[k + 1][0] for k in range(len(rng) - 1))
        nt, pos, rest = len(ready), 0, []
        for lo, hi in sorted(rng) + [(nt, nt)]:                   # the loop of TrainGraph.step
            rest += list(range(pos, lo))
            pos = hi
        assert sorted(rest + [i for lo, hi in rng for i in range(lo, hi)]) == list(range(nt))


def test_close_then_step_does_not_rescale_the_l2_table():
    """TrainGraph.close() detaches the data-parallel reducer and a later step() re-attaches it, but the
    --regularization_penalty scaling of the per-variable l2 table must happen exactly once per TrainGraph."""
    class Reducer(object):
        attached = detached = broadcasts = 0

        def attach(self, g, broadcast=True):
            self.attached += 1
            self.broadcasts += int(bool(broadcast))

        def detach(self):
            self.detached += 1

    class TG(object):
        ensure_finalized = train.TrainGraph.ensure_finalized
        close = train.TrainGraph.close

    tg = TG()
    tg.graph = Graph(device="cpu", seed=0)
    tg.graph.begin_step()
    tg.graph.get_variable("gates/weights", (6, 9), random_normal(0.3), l2=1e-8)
    tg.reg_penalty, tg.reducer = 3.0, Reducer()
    tg.ensure_finalized()
    l2 = tg.graph.l2.clone()
    assert float(l2.max()) == pytest.approx(3e-8) and tg.reducer.attached == 1
    for cycle in range(3):
        tg.ensure_finalized()                                   # idempotent while attached
        tg.close()
        tg.close()                                              # idempotent while detached
        tg.ensure_finalized()
        assert torch.equal(tg.graph.l2, l2)
    assert tg.reducer.attached == 4 and tg.reducer.detached == 3
    assert tg.reducer.broadcasts == 1          # a re-attach must not broadcast between a step's forward and backward pass


def test_auto_cu_reserve_rule_of_the_data_parallel_reducer():
    """the CU headroom of the persistent recurrences is chosen by code when world > 1 (parallel.auto_reserve_cus): nothing
    at world 1 or without recurrent variables; at world 8 the head's 386 MB bucket at the assumed 300 GB/s bus bandwidth is on the wire
    2.25 ms, 0.35 ms past the lone first recurrence -- cheaper than chaining every pair of recurrences (2.7 ms): no reserve; a slow
    fabric (50 GB/s) leaves 11.6 ms of collision: reserve RCCL's CUs."""
    import yt8m_amd.parallel as parallel
    head = 96_500_000 * 4
    assert parallel.auto_reserve_cus(1, head) == 0
    assert parallel.auto_reserve_cus(8, head, persistent_recurrences=False) == 0
    assert parallel.allreduce_ms(head, 8, 300.0) == pytest.approx(2.2517, rel=1e-3)
    assert parallel.auto_reserve_cus(8, head) == 0
    slow = parallel.DP_BUSBW_GBPS
    try:
        parallel.DP_BUSBW_GBPS = 50.0
        assert parallel.auto_reserve_cus(8, head) == parallel.DP_RCCL_CUS
    finally:
        parallel.DP_BUSBW_GBPS = slow
    # the reducer resolves "auto" at attach(): a graph without recurrent variables gets 0 and the rule is recorded
    g = Graph(device="cpu", seed=0)
    g.begin_step()
    g.get_variable("gates/weights", (6, 9), random_normal(0.3), l2=1e-8)
    g.finalize()
    red = parallel.GradReducer(reserve_cus=None)
    if parallel.DP_RESERVED_CUS is None:
        red.attach(g)
        assert red.reserve_cus == 0 and red.reserve_rule["rule"] == "auto"


def test_host_rules_of_the_byte_reading_plugins_and_the_hoisted_h2_roles():
    """Host-side decisions added in the last session of round 6 (no GPU work): which recurrent layers' hoisted projections declare the h2
    role (ops._hoisted_role: the fully-connected layers' size rule, never in the bf16 configuration), and that the byte paths of the
    plugins refuse what their kernels do not cover (seq_ops.u8_cnn_supported / u8_attention_supported: device tensors of uint8 frames
    only -- a CPU tensor falls back to the dequantised float path instead of reaching the library)."""
    import yt8m_amd.ops as ops
    import yt8m_amd.seq_ops as seq_ops
    assert ops._hoisted_role(38400, 4096, 1152) == "h2" and ops._hoisted_role(38400, 2048, 1024) == "h2"
    assert ops._hoisted_role(38400, 4096, 256) is None          # short reduction: not worth the operand passes
    assert ops._hoisted_role(300, 4094, 1152) is None           # N % 4 != 0: the scaled epilogue stores float4
    assert ops._hoisted_role(64, 64, 512) is None               # small layer
    assert ops._hoisted_role(38400, 4096, 1152, bf16=True) is None
    q = torch.zeros((16, 4, 32), dtype=torch.uint8)
    assert not seq_ops.u8_cnn_supported(q) and not seq_ops.u8_attention_supported(q, 8)
    assert not seq_ops.u8_cnn_supported(torch.zeros((16, 4, 32)))
    w = ops._RowWindow(torch.zeros((4, 8)))
    assert tuple(w.data.shape) == (4, 8)
    import yt8m_amd.frame_level_models as flm
    for cls in (flm.LstmModel, flm.LstmMemoryModel, flm.LstmAttentionMaxPoolingModel, flm.LstmPositionalAttentionMaxPoolingModel,
                flm.LstmParallelFinaloutputModel, flm.CnnDeepCombineChainModel, flm.DbofModel, flm.NetVLADModel, flm.GruPoolingModel,
                flm.GruWithPoolingModel, flm.LayerNormLstmMemoryModel, flm.FrameLevelLogisticModel):
        assert getattr(cls, "accepts_quantized_input", False), cls.__name__
    assert not seq_ops.u8_hoisted_supported(q)
