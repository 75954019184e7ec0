"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/yt8m_hip.h declares, and rejects bad arguments with status codes (no compute, no GPU needed);
the Python plugin surface behaves like the reference's (names, lookup, error types)."""
import ctypes
import os
import re

import pytest
import torch

import yt8m_amd._lib as L
from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "yt8m_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(yt8m_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    names = _declared()
    assert len(names) >= 30
    assert set(names) == set(L.SIGNATURES), (set(names) ^ set(L.SIGNATURES))


def test_library_exports_every_declared_symbol():
    lib = L.lib()
    raw = ctypes.CDLL(L.LIB_PATH)
    for n in _declared():
        assert hasattr(raw, n), n
    assert lib.yt8m_abi_version() == 4
    assert lib.yt8m_built_arch() == b"gfx950"


def test_library_exports_nothing_the_header_does_not_declare():
    """The reverse direction: every yt8m_* symbol the shared library exports is declared in include/yt8m_hip.h (an entry point a
    host can bind but cannot find documented would be a boundary leak).  Needs binutils' nm; skipped without it."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or shutil.which("llvm-nm") or ("/opt/rocm/lib/llvm/bin/llvm-nm" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-nm") else None)
    if nm is None:
        pytest.skip("no nm in this environment")
    out = subprocess.run([nm, "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("yt8m_")})
    assert exported == _declared(), sorted(set(exported) ^ set(_declared()))


def test_argument_validation_without_device():
    lib = L.lib()
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    assert lib.yt8m_gemm_f32(0, 0, 4, 4, 4, one, 4, one, 4, one, 4, None, 0.5, None) == -1   # beta not 0/1
    assert b"beta" in lib.yt8m_last_error()
    assert lib.yt8m_gemm_f32(0, 0, -1, 4, 4, one, 4, one, 4, one, 4, None, 0.0, None) == -2   # negative dim
    assert lib.yt8m_gemm_f32(0, 0, 4, 4, 4, one, 2, one, 4, one, 4, None, 0.0, None) == -2    # lda < K
    assert lib.yt8m_gemm_f32(0, 0, 4, 4, 4, None, 4, one, 4, one, 4, None, 0.0, None) == -1   # null operand
    assert lib.yt8m_gemm_f32(0, 0, 0, 4, 4, None, 4, None, 4, None, 4, None, 0.0, None) == 0  # empty problem is a no-op
    assert lib.yt8m_moe_mix_fwd(one, one, one, 2, 3, 0, None) == -1                            # M out of range
    assert lib.yt8m_moe_mix_fwd(one, one, one, 2, 3, 17, None) == -1
    assert lib.yt8m_moe_mix_fwd(None, None, None, 0, 3, 2, None) == 0
    assert lib.yt8m_xent_fwd_bwd(one, one, 0, None, one, None, 0, 5, 1e-5, 1.0, one, None) == -2  # empty batch
    assert lib.yt8m_xent_fwd_bwd(one, one, 7, None, one, None, 2, 5, 1e-5, 1.0, one, None) == -1  # label dtype
    assert lib.yt8m_topk_rows(one, 2, 10, 0, one, one, None) == -1
    assert lib.yt8m_topk_rows(one, 2, 10, 11, one, one, None) == -1
    assert lib.yt8m_act_fwd_f32(9, one, one, 4, None) == -1
    assert lib.yt8m_xent_workspace_bytes(1024, 4716) == 4 * (1024 * 5 + 1)
    # entry points added for the recurrent cells, the random ops, the streaming FC / pooling kernels and the bf16 casts
    assert lib.yt8m_dropout_f32(one, one, 8, 0.0, 1, 0, None) == -1                             # keep_prob out of (0, 1]
    assert lib.yt8m_dropout_f32(one, one, 8, 1.5, 1, 0, None) == -1
    assert lib.yt8m_dropout_f32(one, one, -1, 0.5, 1, 0, None) == -2
    assert lib.yt8m_dropout_f32(None, None, 0, 0.5, 1, 0, None) == 0
    assert lib.yt8m_add_noise_f32(one, one, 8, -1.0, 1, 0, None) == -1
    assert lib.yt8m_skinny_supported(100, 1152, 8) == 1 and lib.yt8m_skinny_supported(100, 1152, 17) == 0
    assert lib.yt8m_skinny_supported(100, 1150, 8) == 0 and lib.yt8m_skinny_supported(100, 8192, 8) == 0
    assert lib.yt8m_skinny_fwd_f32(one, 8, one, 17, None, one, 17, 4, 8, 17, 0.0, None) == -2   # N > 16
    assert lib.yt8m_skinny_fwd_f32(one, 6, one, 4, None, one, 4, 4, 6, 4, 0.0, None) == -2      # K % 4 != 0
    assert lib.yt8m_skinny_fwd_f32(one, 8, one, 4, None, one, 4, 4, 8, 4, 0.5, None) == -1      # beta
    assert lib.yt8m_skinny_dw_f32(one, 8, one, 4, one, 4, 4, 8, 4, 0.0, None, 0, None) == -1    # no workspace
    assert lib.yt8m_skinny_workspace_bytes(307200, 1152, 8) >= 64 * 1152 * 8 * 4
    assert lib.yt8m_attn_pool_supported(128, 300, 8, 1152) == 1 and lib.yt8m_attn_pool_supported(128, 300, 17, 1152) == 0
    assert lib.yt8m_attn_pool_fwd(one, one, one, 2, 10, 17, 64, None) == -2
    assert lib.yt8m_attn_pool_fwd(None, None, None, 0, 10, 8, 64, None) == 0
    assert lib.yt8m_cast_f32_bf16(one, 4, 8, 8, one, 4, 0, None) == -2                          # dst_ld < cols
    assert lib.yt8m_cast_f32_bf16_dual(one, 4, 8, 8, one, 8, one, 2, None) == -2                # trans_ld < rows
    assert lib.yt8m_moe_mix_bwd_bf16(one, one, one, None, 0, 4, 8, 3, 1e-5, 1.0, None, one, 24, one, 8, one, 16, one, 8, None, None) == -1   # M != 2
    assert lib.yt8m_moe_mix_bwd_bf16(one, one, one, one, 0, 4, 8, 2, 1e-5, 1.0, None, one, 24, one, 8, one, 16, one, 8, None, None) == -1   # dp AND labels
    assert lib.yt8m_moe_mix_bwd_bf16(one, one, one, None, 0, 4, 8, 2, 1e-5, 1.0, None, one, 20, one, 8, one, 16, one, 8, None, None) == -2   # pitch < 3V
    assert lib.yt8m_moe_mix_bwd_bf16_partial_rows(1024) == 16 and lib.yt8m_moe_mix_bwd_bf16_partial_rows(65) == 2
    assert lib.yt8m_lstm_packed16_elems(128, 1024) == 1024 * 4096 and lib.yt8m_lstm_packed16_elems(128, 384) == 0
    assert lib.yt8m_lstm_pack_bf16(one, 4 * 384, 384, one, None, None) == -2                    # H % 256 != 0
    assert lib.yt8m_lstm_steps_fwd_bf16(one, one, one, one, None, None, None, 0, 2, 2, 256, 1.0, None) == -1   # hs16 missing
    assert lib.yt8m_lstm_steps_bwd_bf16(one, one, one, None, one, one, one, 2, None, 0, 2, 2, 256, None) == -1  # phase
    assert lib.yt8m_gru_layer_fwd(None, one, one, 8, one, 4, one, one, None, None, 2, 2, 4, None, 0, None) == -1
    assert lib.yt8m_gru_layer_fwd(one, one, one, 4, one, 4, one, one, None, None, 2, 2, 4, None, 0, None) == -2   # ldg < 2H
    assert lib.yt8m_lnlstm_layer_fwd(one, one, 16, one, one, one, one, one, None, None, 2, 2, 4096, 1.0, 1.0, 0, None, 0, None) == -2
    assert lib.yt8m_lnlstm_layer_fwd(one, one, 16, one, one, one, one, one, None, None, 2, 2, 4, 1.0, 0.0, 0, None, 0, None) == -1
    with pytest.raises(ValueError):
        L.check(-2)
    with pytest.raises(L.Yt8mHipError):
        L.check(-3)


def test_ops_fail_loudly_on_host_tensors():
    """No CPU fallback: a host tensor is an error, never a silent eager path."""
    import yt8m_amd.ops as ops
    a = torch.zeros(4, 4)
    with pytest.raises(L.Yt8mHipError):
        ops.gemm(a, a)
    with pytest.raises(L.Yt8mHipError):
        ops.l2norm_fwd(a)
    with pytest.raises(L.Yt8mHipError):
        ops.xent_fwd(a, a)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.Yt8mHipError, match="no CPU fallback"):
        L.lib()


def test_plugin_surface(flags):
    import yt8m_amd.models as models
    import yt8m_amd.video_level_models as vlm
    import yt8m_amd.frame_level_models as flm
    import yt8m_amd.losses as losses
    import yt8m_amd.train as train
    with pytest.raises(NotImplementedError):
        models.BaseModel().create_model(None)
    with pytest.raises(NotImplementedError):
        losses.BaseLoss().calculate_loss(None, None)
    # W/train.py:212-215,703-708
    assert train.find_class_by_name("MoeModel", [flm, vlm]) is vlm.MoeModel
    assert train.find_class_by_name("LstmModel", [flm, vlm]) is flm.LstmModel
    assert train.find_class_by_name("CrossEntropyLoss", [losses]) is losses.CrossEntropyLoss
    with pytest.raises(StopIteration):
        train.find_class_by_name("NoSuchModel", [flm, vlm])
    for name in ["LogisticModel", "MoeModel", "DeepCombineChainModel"]:
        assert issubclass(getattr(vlm, name), models.BaseModel)
    for name in ["LstmModel", "LstmMemoryModel", "LstmAttentionMaxPoolingModel", "DbofModel", "FrameLevelLogisticModel",
                 "NetVLADModel", "GatedNetVLADModel",
                 "GatedNetVLADAttentionChainModel", "LstmParallelFinaloutputModel", "LstmPositionalAttentionMaxPoolingModel",
                 "CnnDeepCombineChainModel", "GruPoolingModel", "GruWithPoolingModel", "LayerNormLstmMemoryModel"]:
        assert issubclass(getattr(flm, name), models.BaseModel), name
    # reference flag names / defaults
    assert flags.moe_num_mixtures == 2 and flags.deep_chain_layers == 3 and flags.deep_chain_relu_cells == 200
    assert flags.lstm_cells == "1024" and flags.lstm_layers == 2 and flags.lstm_attentions == 8
    assert flags.gru_cells == 1024 and flags.gru_layers == 2 and flags.dropout is False and flags.keep_prob == 1.0
    assert flags.noise_level == 0.0
    assert flags.video_level_classifier_model == "MoeModel" and flags.dbof_cluster_size == 8192
    assert flags.batch_size == 1024 and flags.base_learning_rate == 0.01 and flags.clip_gradient_norm == 1.0
    assert flags.learning_rate_decay == 0.95 and flags.learning_rate_decay_examples == 4000000
    assert flags.label_loss == "CrossEntropyLoss" and flags.num_classes == 4716 and flags.support_loss_percent == 0.1
    rest = flags.parse(["--moe_num_mixtures=4", "--lstm_cells", "512", "--multitask", "x.txt", "--nolabel_smoothing"])
    assert flags.moe_num_mixtures == 4 and flags.lstm_cells == "512" and flags.multitask is True and rest == ["x.txt"]
    with pytest.raises(ValueError):
        flags.parse(["--no_such_flag=1"])


def test_lr_schedule_and_variable_store(flags):
    import yt8m_amd.train as train
    from yt8m_amd.variables import Graph, xavier_uniform, zeros, CHUNK
    assert train.exponential_decay(0.01, 3906, 1024, 4000000, 0.95) == 0.01
    assert train.exponential_decay(0.01, 3907, 1024, 4000000, 0.95) == pytest.approx(0.0095)
    g = Graph(device="cpu", seed=1)
    with g.variable_scope("RNN"):
        w = g.get_variable("cell/weights", (5000, 3), xavier_uniform, l2=1e-8)
    b = g.get_variable("experts/biases", (7,), zeros)
    assert w.name == "RNN/cell/weights" and g.get_variable("experts/biases", (7,)) is b
    with pytest.raises(ValueError):
        g.get_variable("experts/biases", (8,))
    g.begin_step()
    a0 = g.anonymous_variable((2, 2), zeros)
    a1 = g.anonymous_variable((3,), zeros)
    assert (a0.name, a1.name) == ("Variable", "Variable_1")
    g.begin_step()
    assert g.anonymous_variable((2, 2), zeros) is a0
    before = w.data.clone()
    g.finalize()
    assert torch.equal(w.data, before) and w.data.data_ptr() == g.params.data_ptr()
    assert w.offset == 0 and b.offset % 64 == 0 and b.offset >= 15000
    ch = g.chunks.tolist()
    assert g.nchunks == 4 + 1 + 1 + 1 and ch[0] == [0, CHUNK, 0, 0] and ch[3] == [3 * CHUNK, 15000 - 3 * CHUNK, 0, 0]
    assert ch[4] == [b.offset, 7, 1, 0] and g.l2.tolist() == pytest.approx([1e-8, 0, 0, 0])
    assert w.grad.shape == (5000, 3) and w.grad_beta() == 0.0 and w.grad_beta() == 1.0
    g.begin_step()
    assert w.grad_beta() == 0.0
    with pytest.raises(RuntimeError):
        g.get_variable("late", (1,))
    sd = g.state_dict()
    assert set(sd) == {"RNN/cell/weights", "experts/biases", "Variable", "Variable_1"}
    lim = (6.0 / (5000 + 3)) ** 0.5
    assert float(w.data.abs().max()) <= lim and float(w.data.abs().max()) > 0.9 * lim


def test_comm_entry_points_fail_loudly_without_a_gpu():
    """yt8m_comm_*: argument errors are YT8M_E_BADARG; anything RCCL refuses (no ROCm device on this box) is YT8M_E_RCCL with
    the RCCL message in yt8m_last_error -- never a silent fallback."""
    import ctypes
    import yt8m_amd._lib as L
    lib = L.lib()
    buf = ctypes.create_string_buffer(128)
    comm = ctypes.c_void_p()
    assert lib.yt8m_comm_unique_id(None) == -1
    assert lib.yt8m_comm_init(0, 0, buf, ctypes.byref(comm)) == -1 and lib.yt8m_comm_init(2, 2, buf, ctypes.byref(comm)) == -1
    assert lib.yt8m_comm_allreduce_mean(None, None, 4, None) == -1
    assert lib.yt8m_comm_destroy(None) == 0
    rc = lib.yt8m_comm_unique_id(buf)
    assert rc in (0, -4)
    if rc == -4:
        assert b"ncclGetUniqueId" in lib.yt8m_last_error() or b"RCCL" in lib.yt8m_last_error()
