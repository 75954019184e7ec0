"""ctypes binding of libyt8m_hip.so (include/yt8m_hip.h).  Fails loudly when the library is missing:
the product has no CPU / eager-PyTorch fallback for any hot op."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YT8M_LIB", os.path.join(_HERE, "libyt8m_hip.so"))  # YT8M_LIB: A/B builds (tools/)

c_void_p, c_int, c_int64, c_float, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double
P = c_void_p

class GemmProblem(ctypes.Structure):
    """yt8m_gemm_problem (include/yt8m_hip.h)."""
    _fields_ = [("M", c_int64), ("N", c_int64), ("K", c_int64), ("A", c_void_p), ("lda", c_int64), ("B", c_void_p),
                ("ldb", c_int64), ("C", c_void_p), ("ldc", c_int64), ("bias", c_void_p), ("beta", c_float)]


STREAM_HOOK = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p)       # yt8m_stream_hook


class PersistBwdImages(ctypes.Structure):
    """yt8m_persist_bwd_images (include/yt8m_hip.h): where yt8m_lstm_persist_bwd_images leaves the operand images of its dz."""
    _fields_ = [("plain", ctypes.c_void_p), ("trans", ctypes.c_void_p), ("trans_scaled", ctypes.c_void_p), ("rowscale", ctypes.c_void_p),
                ("colpart", ctypes.c_void_p), ("colpart_scaled", ctypes.c_void_p)]


class LstmStackDesc(ctypes.Structure):
    """yt8m_lstm_stack_desc (include/yt8m_hip.h)."""
    _fields_ = [("B", c_int64), ("F", c_int64), ("D", c_int64), ("H", c_int64), ("L", ctypes.c_int32), ("input_u8", ctypes.c_int32),
                ("forget_bias", c_float), ("fwd_chunks", ctypes.c_int32), ("bwd_chunks", ctypes.c_int32), ("need_dx", ctypes.c_int32),
                ("input_keep_prob", c_float), ("reserved0", ctypes.c_int32), ("dropout_seed", ctypes.c_uint64 * 8)]      # ABI 4


class WimgDemand(ctypes.Structure):
    """yt8m_wimg_demand (include/yt8m_hip.h)."""
    _fields_ = [("src", c_void_p), ("R", c_int64), ("C", c_int64), ("ld", c_int64), ("trans", ctypes.c_int32), ("planes", ctypes.c_int32),
                ("scale", c_float), ("pad", ctypes.c_int32)]


class WimgSpec(ctypes.Structure):
    """yt8m_wimg_spec (include/yt8m_hip.h)."""
    _fields_ = [("plain", c_void_p), ("trans", c_void_p), ("row0", c_int64), ("rows", c_int64), ("scale", c_float),
                ("planes", ctypes.c_int32)]


class WimgJob(ctypes.Structure):
    """yt8m_wimg_job (include/yt8m_hip.h)."""
    _fields_ = [("offset", c_int64), ("R", c_int64), ("C", c_int64), ("tensor", ctypes.c_int32), ("nspec", ctypes.c_int32),
                ("tile_base", c_int64), ("spec", WimgSpec * 4)]


class OptRanges(ctypes.Structure):
    """yt8m_opt_ranges (include/yt8m_hip.h)."""
    _fields_ = [("w", c_void_p), ("m", c_void_p), ("v", c_void_p), ("g", c_void_p), ("chunks", c_void_p), ("tensor_chunk_start", c_void_p),
                ("tensor_chunk_start_host", c_void_p), ("l2", c_void_p), ("partial", c_void_p), ("norms", c_void_p), ("skip_tensor", c_void_p),
                ("jobs", c_void_p), ("job_tensor_host", c_void_p), ("job_tile_base_host", c_void_p), ("njobs", ctypes.c_int32),
                ("nranges", ctypes.c_int32), ("range_lo", ctypes.c_int32 * 8), ("range_hi", ctypes.c_int32 * 8), ("gscale", c_float),
                ("clip", c_float), ("lr_t", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("after_stream", c_void_p)]


DESC = ctypes.POINTER(LstmStackDesc)
PP = ctypes.POINTER(c_void_p)      # array of device pointers

# name -> (restype, argtypes); one row per function declared in include/yt8m_hip.h
SIGNATURES = {
    "yt8m_abi_version": (c_int, []),
    "yt8m_last_error": (ctypes.c_char_p, []),
    "yt8m_built_arch": (ctypes.c_char_p, []),
    "yt8m_prof_enable": (c_int, [c_int]),
    "yt8m_prof_reset": (c_int, []),
    "yt8m_prof_get": (c_int, [c_int, ctypes.POINTER(c_int64), ctypes.POINTER(c_double)]),
    "yt8m_prof_get_flops": (c_int, [c_int, ctypes.POINTER(c_double)]),
    "yt8m_prof_get_bytes": (c_int, [c_int, ctypes.POINTER(ctypes.c_double)]),
    "yt8m_probe_mfma_f32": (c_int, [c_int, c_int, P, P]),
    "yt8m_probe_mfma_bf16": (c_int, [c_int, c_int, c_int, P, P]),
    "yt8m_probe_copy_f32": (c_int, [P, P, c_int64, P]),
    "yt8m_probe_placement": (c_int, [P, c_int, c_int, P]),
    "yt8m_stream_create_cu_mask": (c_int, [P, c_int, ctypes.POINTER(P)]),
    "yt8m_stream_destroy": (c_int, [P]),
    "yt8m_gemm_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_float, P]),
    "yt8m_gemm_workspace_bytes": (c_int64, []),
    "yt8m_gemm_f32_grouped": (c_int, [c_int, c_int, c_int, ctypes.POINTER(GemmProblem), P, c_int64, P]),
    "yt8m_gemm_bf16_nt_grouped": (c_int, [c_int, ctypes.POINTER(GemmProblem), P, c_int64, P]),
    "yt8m_gemm_x3_pays": (c_int, [c_int64, c_int64, c_int64]),
    "yt8m_gemm_auto_scratch_bytes": (c_int64, [c_int, c_int, c_int, ctypes.POINTER(GemmProblem)]),
    "yt8m_gemm_auto_grouped": (c_int, [c_int, c_int, c_int, ctypes.POINTER(GemmProblem), P, c_int64, P, c_int64, ctypes.POINTER(ctypes.c_uint64), P]),
    "yt8m_gemm_auto_grouped_ex": (c_int, [c_int, c_int, c_int, ctypes.POINTER(GemmProblem), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                          P, c_int64, P, c_int64, ctypes.POINTER(ctypes.c_uint64), P]),
    "yt8m_x3_image_bytes": (c_int64, [c_int64, c_int64]),
    "yt8m_x3_split": (c_int, [P, c_int64, c_int64, c_int64, c_float, P, P, P]),
    "yt8m_gemm_x3_nt_grouped": (c_int, [c_int, ctypes.POINTER(GemmProblem), P, c_int64, P]),
    "yt8m_gemm_x1x3_nt": (c_int, [c_int64, c_int64, c_int64, P, P, P, c_int64, P, P, P, c_float, P, c_int64, P]),
    "yt8m_gemm_x1x3_nt_ex": (c_int, [c_int64, c_int64, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_float, P, P, c_float, c_float, P,
                                     c_int64, P]),
    "yt8m_bf16_image": (c_int, [P, c_int64, c_int64, c_int64, c_float, P, P, P]),
    "yt8m_gemm_b1_nt_grouped": (c_int, [c_int, ctypes.POINTER(GemmProblem), P, c_int64, P]),
    "yt8m_gemm_b1_nt_grouped_bf16c": (c_int, [c_int, ctypes.POINTER(GemmProblem), ctypes.c_uint, P, c_int64, P]),
    "yt8m_x3_split_ex": (c_int, [P, c_int64, c_int64, c_int64, c_float, P, P, P, P, P]),
    "yt8m_x3_split_colsum": (c_int, [P, c_int64, c_int64, c_int64, c_float, P, P, P, P, P, P, P]),
    "yt8m_x3_set_combine": (c_int, [c_int]),
    "yt8m_x3_set_schedule": (c_int, [c_int]),
    "yt8m_bf16_image_colsum": (c_int, [P, c_int64, c_int64, c_int64, c_float, P, P, P, P, P, P, P]),
    "yt8m_gemm_b1_nt_ex": (c_int, [c_int64, c_int64, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_float, P, P, c_float, c_float, P,
                                   c_int64, P]),
    "yt8m_u8_frames_image_t": (c_int, [P, P, c_int64, c_int64, c_int64, P, P]),
    "yt8m_lstm_stack_supported": (c_int, [DESC]),
    "yt8m_lstm_stack_tape_bytes": (c_int64, [DESC]),
    "yt8m_lstm_stack_scratch_bytes": (c_int64, [DESC]),
    "yt8m_lstm_stack_partition": (c_int, [DESC, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "yt8m_lstm_stack_set_prep_hook": (c_int, [P, P]),
    "yt8m_lstm_stack_streams": (c_int, [c_int, PP, ctypes.POINTER(c_void_p)]),
    "yt8m_lstm_stack_fwd": (c_int, [DESC, P, P, PP, PP, P, c_int64, P, c_int64, P]),
    "yt8m_lstm_stack_view": (c_int, [DESC, P, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "yt8m_lstm_stack_bwd": (c_int, [DESC, P, P, PP, P, c_int64, P, c_int64, P, PP, PP, PP, PP, ctypes.POINTER(c_float),
                                    ctypes.POINTER(c_float), P, P]),
    "yt8m_lstm_stack_status": (c_int, [DESC, P, P]),
    "yt8m_lstm_stack_layer_done_wait": (c_int, [c_int, P]),
    "yt8m_u8_frames_image": (c_int, [P, P, c_int64, c_int64, c_int64, c_float, P, P, P, P]),
    "yt8m_cast_f32_bf16": (c_int, [P, c_int64, c_int64, c_int64, P, c_int64, c_int, P]),
    "yt8m_cast_f32_bf16_dual": (c_int, [P, c_int64, c_int64, c_int64, P, c_int64, P, c_int64, P]),
    "yt8m_gemm_f32_batched": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, P, c_int64, c_int64, P, c_int64, c_int64,
                                      P, c_int64, c_int64, c_float, c_int64, P]),
    "yt8m_l2norm_fwd_f32": (c_int, [P, P, c_int64, c_int64, c_float, P]),
    "yt8m_l2norm_bwd_f32": (c_int, [P, P, P, c_int64, c_int64, c_float, P]),
    "yt8m_dequant_l2norm_u8": (c_int, [P, P, P, c_int64, c_int64, c_int64, c_float, P]),
    "yt8m_dequant_mean_l2norm_u8": (c_int, [P, P, P, c_int64, c_int64, c_int64, c_float, P]),
    "yt8m_moe_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "yt8m_moe_workspace_bytes_ex": (c_int64, [c_int64, c_int64, c_int64, c_int]),
    "yt8m_moe_fwd": (c_int, [P, P, P, P, P, c_int, c_int64, c_int64, c_int64, c_int, c_float, P, P, P, P, P, c_int64, P]),
    "yt8m_moe_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int64, c_int64, c_int64, c_int, c_float, c_float, P, P, P, c_float, P, P,
                             c_int64, P]),
    "yt8m_logistic_fwd_bwd": (c_int, [P, P, P, P, c_int, c_int64, c_int64, c_int64, c_float, P, P, P, P, P, c_float, P, P,
                                      c_int64, P]),
    "yt8m_moe_mix_fwd": (c_int, [P, P, P, c_int64, c_int64, c_int, P]),
    "yt8m_moe_mix_bwd": (c_int, [P, P, P, c_int64, c_int64, c_int, P]),
    "yt8m_moe_mix_xent_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "yt8m_moe_mix_xent_fwd": (c_int, [P, P, P, c_int, P, P, c_int64, c_int64, c_int, c_float, P, P]),
    "yt8m_moe_mix_xent_bwd": (c_int, [P, P, P, c_int, P, c_int64, c_int64, c_int, c_float, c_float, P]),
    "yt8m_moe_mix_xent_bwd_absmax": (c_int, [P, P, P, c_int, P, c_int64, c_int64, c_int, c_float, c_float, P, P]),
    "yt8m_skinny_supported": (c_int, [c_int64, c_int64, c_int64]),
    "yt8m_skinny_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "yt8m_skinny_fwd_f32": (c_int, [P, c_int64, P, c_int64, P, P, c_int64, c_int64, c_int64, c_int64, c_float, P]),
    "yt8m_skinny_dw_f32": (c_int, [P, c_int64, P, c_int64, P, c_int64, c_int64, c_int64, c_int64, c_float, P, c_int64, P]),
    "yt8m_skinny_dx_f32": (c_int, [P, c_int64, P, c_int64, P, c_int64, c_int64, c_int64, c_int64, c_float, P]),
    "yt8m_lstm_packed16_elems": (c_int64, [c_int64, c_int64]),
    "yt8m_lstm_pack_bf16": (c_int, [P, c_int64, c_int64, P, P, P]),
    "yt8m_lstm_steps_fwd_bf16": (c_int, [P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, P]),
    "yt8m_lstm_steps_bwd_bf16": (c_int, [P, P, P, P, P, P, P, c_int, P, c_int64, c_int64, c_int64, c_int64, P]),
    "yt8m_u8_frame_scales": (c_int, [P, P, c_int64, c_int64, c_int64, c_float, P, P]),
    "yt8m_skinny_fwd_u8": (c_int, [P, c_int64, P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, P]),
    "yt8m_skinny_dw_u8": (c_int, [P, c_int64, P, c_int64, P, P, c_int64, c_int64, c_int64, c_int64, c_float, P, c_int64, P]),
    "yt8m_attn_pool_fwd_u8": (c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int64, P]),
    "yt8m_attn_pool_dw_u8": (c_int, [P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, P]),
    "yt8m_attn_pool_supported": (c_int, [c_int64, c_int64, c_int64, c_int64]),
    "yt8m_attn_pool_fwd": (c_int, [P, P, P, c_int64, c_int64, c_int64, c_int64, P]),
    "yt8m_attn_pool_bwd": (c_int, [P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, P]),
    "yt8m_gru_layer_fwd": (c_int, [P, P, P, c_int64, P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, P, c_int64, P]),
    "yt8m_gru_layer_bwd": (c_int, [P, P, P, c_int64, P, c_int64, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, P, c_int64, P]),
    "yt8m_lnlstm_layer_fwd": (c_int, [P, P, c_int64, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_float, c_float,
                                      ctypes.c_uint64, P, c_int64, P]),
    "yt8m_lnlstm_layer_bwd": (c_int, [P, P, c_int64, P, P, P, P, P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_float,
                                      c_float, ctypes.c_uint64, P, c_int64, P]),
    "yt8m_dropout_f32": (c_int, [P, P, c_int64, c_float, ctypes.c_uint64, c_int64, P]),
    "yt8m_add_noise_f32": (c_int, [P, P, c_int64, c_float, ctypes.c_uint64, c_int64, P]),
    "yt8m_moe_mix_bwd_bf16_partial_rows": (c_int64, [c_int64]),
    "yt8m_moe_mix_bwd_bf16": (c_int, [P, P, P, P, c_int, c_int64, c_int64, c_int, c_float, c_float, P, P, c_int64, P, c_int64,
                                      P, c_int64, P, c_int64, P, P]),
    "yt8m_moe_mix_bwd_bf16_images": (c_int, [P, P, P, P, c_int, c_int64, c_int64, c_int, c_float, c_float, P, P, c_int64, P, c_int64,
                                      P, c_int64, P, c_int64, P, P]),
    "yt8m_moe_mix_bwd_bf16_images_z16": (c_int, [P, P, P, P, c_int, c_int64, c_int64, c_int, c_float, c_float, P, P, c_int64, P, c_int64,
                                      P, c_int64, P, c_int64, P, P]),
    "yt8m_moe_mix_fwd_bf16z": (c_int, [P, P, P, c_int64, c_int64, c_int, P]),
    "yt8m_act_fwd_f32": (c_int, [c_int, P, P, c_int64, P]),
    "yt8m_act_bwd_f32": (c_int, [c_int, P, P, P, c_int64, P]),
    "yt8m_colsum_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "yt8m_colsum_f32": (c_int, [P, c_int64, c_int64, c_int64, P, c_float, P, c_int64, P]),
    "yt8m_rank1_add_rows_f32": (c_int, [P, c_int64, c_int64, c_int64, P, c_float, P]),
    "yt8m_colsum_weighted_f32": (c_int, [P, c_int64, c_int64, c_int64, P, P, c_float, P, P, c_int64, P]),
    "yt8m_xent_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "yt8m_xent_fwd_bwd": (c_int, [P, P, c_int, P, P, P, c_int64, c_int64, c_float, c_float, P, P]),
    "yt8m_xent_bwd": (c_int, [P, P, c_int, P, P, P, c_int64, c_int64, c_float, c_float, P]),
    "yt8m_sqnorm_multi": (c_int, [P, P, P, c_int64, P, c_float, P, P, c_int64, c_int64, P, c_int64, P]),
    "yt8m_adam_multi": (c_int, [P, P, P, P, P, c_int64, P, c_float, P, c_float, c_float, c_float, c_float, c_float, P]),
    "yt8m_lstm_gates_fwd": (c_int, [P, P, P, P, P, P, P, ctypes.c_int32, c_int64, c_int64, c_float, P]),
    "yt8m_lstm_gates_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, ctypes.c_int32, c_int64, c_int64, P]),
    "yt8m_lstm_layer_fwd": (c_int, [P, P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_float, P, c_int64, P]),
    "yt8m_graph_cache_stats": (c_int, [ctypes.POINTER(c_int64), ctypes.POINTER(c_int64), ctypes.POINTER(c_int64),
                                       ctypes.POINTER(c_int64)]),
    "yt8m_graph_cache_clear": (c_int, []),
    "yt8m_lstm_persist_supported": (c_int, [c_int64, c_int64]),
    "yt8m_lstm_persist_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "yt8m_lstm_persist_workspace_bytes_steps": (c_int64, [c_int64, c_int64, c_int64]),
    "yt8m_lstm_persist_placement_stats": (c_int, [ctypes.POINTER(c_int64), ctypes.POINTER(c_int64), ctypes.POINTER(c_int64), c_int]),
    "yt8m_lstm_persist_fwd_on_bf16_pipe": (c_int, [c_int64, c_int64]),
    "yt8m_lstm_persist_set_cus": (c_int, [c_int, c_int]),
    "yt8m_lstm_persist_status": (c_int, [P, P]),
    "yt8m_lstm_persist_reserve_cus": (c_int, [c_int, ctypes.POINTER(c_int)]),
    "yt8m_lstm_persist_debug_fault": (c_int, [P, P]),
    "yt8m_lstm_persist_fwd": (c_int, [P, P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, P, c_int64, P]),
    "yt8m_lstm_persist_fwd_bf16": (c_int, [P, P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, P, c_int64, P]),
    "yt8m_lstm_persist_bwd_supported": (c_int, [c_int64, c_int64]),
    "yt8m_lstm_persist_bwd": (c_int, [P, P, c_int64, P, P, P, P, c_int, P, P, c_int64, c_int64, c_int64, c_int64, P, c_int64, P]),
    "yt8m_lstm_persist_bwd_bf16": (c_int, [P, P, c_int64, P, P, P, P, c_int, P, P, c_int64, c_int64, c_int64, c_int64, P, c_int64, P]),
    "yt8m_lstm_persist_fwd_h2": (c_int, [P, P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, P, P, c_int64, P]),
    "yt8m_lstm_persist_bwd_h2": (c_int, [P, P, c_int64, P, P, P, P, c_int, P, P, c_int64, c_int64, c_int64, c_int64, P, P, c_int64, P]),
    "yt8m_lstm_persist_bwd_on_f16_pipe": (c_int, [c_int64, c_int64]),
    "yt8m_lstm_persist_bwd_ex": (c_int, [P, P, c_int64, P, P, P, P, c_int, P, c_int64, c_int64, c_int64, c_int64, P, P, P, P, c_int64, P]),
    "yt8m_h2_split_rowmax": (c_int, [P, c_int64, c_int64, c_int64, P, P, P, P]),
    "yt8m_lstm_persist_bwd_images_rows": (c_int, [c_int64, c_int64]),
    "yt8m_lstm_persist_bwd_images": (c_int, [P, P, c_int64, P, P, P, P, c_int, P, c_int64, c_int64, c_int64, c_int64, P, c_int64, P, P]),
    "yt8m_lstm_packed_floats": (c_int64, [c_int64, c_int64]),
    "yt8m_lstm_pack": (c_int, [P, c_int64, c_int64, P, P, P]),
    "yt8m_lstm_steps_fwd": (c_int, [P, P, c_int64, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, P, c_int64, P]),
    "yt8m_lstm_steps_bwd": (c_int, [P, P, c_int64, P, P, P, P, P, c_int, P, c_int64, c_int64, c_int64, c_int64, P, c_int64, P]),
    "yt8m_lstm_layer_bwd": (c_int, [P, P, c_int64, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, P, c_int64, P]),
    "yt8m_attn_softmax_fwd": (c_int, [P, P, P, c_int64, c_int64, c_int64, P]),
    "yt8m_attn_softmax_bwd": (c_int, [P, P, P, P, c_int64, c_int64, c_int64, P]),
    "yt8m_softmax_rows_fwd": (c_int, [P, P, P, c_int64, c_int64, c_int64, P]),
    "yt8m_softmax_rows_bwd": (c_int, [P, P, P, P, c_int64, c_int64, c_int64, P]),
    "yt8m_netvlad_supported": (c_int, [c_int64, c_int64, c_int64, c_int64]),
    "yt8m_netvlad_single_pass": (c_int, [c_int64, c_int64, c_int64, c_int64]),
    "yt8m_netvlad_set_single": (c_int, [c_int]),
    "yt8m_netvlad_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64, c_int64]),
    "yt8m_netvlad_fwd_u8": (c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int, c_float, P, P, P, P, c_int64, P]),
    "yt8m_netvlad_bwd_u8": (c_int, [P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int, c_float, P, c_float, P, c_float,
                                    P, c_int64, P]),
    "yt8m_vlad_finish_fwd": (c_int, [P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, P]),
    "yt8m_vlad_finish_bwd": (c_int, [P, P, P, P, P, P, P, c_float, c_int64, c_int64, c_int64, c_float, P]),
    "yt8m_vlad_finish_q_supported": (c_int, [c_int64]),
    "yt8m_vlad_finish_q_fwd": (c_int, [P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, P]),
    "yt8m_vlad_finish_q_bwd": (c_int, [P, P, P, P, P, P, P, P, c_float, c_int64, c_int64, c_int64, c_float, P]),
    "yt8m_crc32c": (ctypes.c_uint32, [P, c_int64]),
    "yt8m_crc32c_masked": (ctypes.c_uint32, [P, c_int64]),
    "yt8m_prefetch_open": (c_int, [ctypes.POINTER(ctypes.c_char_p), c_int, c_int, ctypes.POINTER(ctypes.c_char_p),
                                   ctypes.POINTER(ctypes.c_int32), c_int, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                                   ctypes.POINTER(c_void_p)]),
    "yt8m_prefetch_acquire": (c_int, [P, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                      ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), ctypes.POINTER(c_int64),
                                      ctypes.POINTER(c_int)]),
    "yt8m_prefetch_close": (c_int, [P]),
    "yt8m_tfrecord_write_predictions": (c_int, [ctypes.c_char_p, c_int64, P, c_int64, P, P, c_int64, ctypes.c_char_p]),
    "yt8m_tfrecord_open": (c_int, [ctypes.c_char_p, c_int, ctypes.POINTER(c_void_p)]),
    "yt8m_tfrecord_close": (c_int, [P]),
    "yt8m_tfrecord_read_frame_batch": (c_int, [P, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int32), c_int, c_int64,
                                               c_int64, c_int64, P, P, P, P, c_int64, ctypes.POINTER(c_int64)]),
    "yt8m_tfrecord_read_video_batch": (c_int, [P, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int32), c_int, c_int64,
                                               c_int64, P, P, P, c_int64, ctypes.POINTER(c_int64)]),
    "yt8m_comm_unique_id": (c_int, [P]),
    "yt8m_comm_init": (c_int, [c_int, c_int, P, ctypes.POINTER(c_void_p)]),
    "yt8m_comm_size": (c_int, [P, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "yt8m_comm_allreduce_f32": (c_int, [P, P, c_int64, c_int, P]),
    "yt8m_comm_allreduce_mean": (c_int, [P, P, c_int64, P]),
    "yt8m_comm_allreduce_rsag_f32": (c_int, [P, P, c_int64, c_int, c_int, P]),
    "yt8m_comm_broadcast_f32": (c_int, [P, P, c_int64, c_int, P]),
    "yt8m_comm_destroy": (c_int, [P]),
    "yt8m_u8_proj_supported": (c_int, [c_int64]),
    "yt8m_u8_frames_to_bf16_tm": (c_int, [P, P, c_int64, c_int64, c_int64, c_float, c_int, P, c_int64, P, P, P]),
    "yt8m_split3_bf16_t": (c_int, [P, c_int64, c_int64, c_int64, c_float, P, c_int64, P]),
    "yt8m_rowscale_bias_f32": (c_int, [P, c_int64, c_int64, c_int64, P, P, c_float, P, P]),
    "yt8m_sample_frames_f32": (c_int, [P, P, c_int64, c_int64, c_int64, c_int64, c_int, ctypes.c_uint64, P, P, P]),
    "yt8m_sample_frames_u8": (c_int, [P, P, c_int64, c_int64, c_int64, c_int64, c_int, ctypes.c_uint64, P, P, P]),
    "yt8m_frame_pool_fwd": (c_int, [P, c_int64, c_int64, c_int64, c_int, P, P]),
    "yt8m_frame_pool_bwd": (c_int, [P, P, P, c_int64, c_int64, c_int64, c_int, P, P]),
    "yt8m_batchnorm_fwd": (c_int, [P, c_int64, c_int64, P, P, P, P, c_int, c_float, c_float, P, P, P, P]),
    "yt8m_batchnorm_workspace_bytes": (c_int64, [c_int64]),
    "yt8m_batchnorm_bwd": (c_int, [P, P, c_int64, c_int64, P, P, P, c_int, P, P, c_float, P, c_float, P, c_int64, P]),
    "yt8m_wimg_register": (c_int, [P, c_int64, c_int64, c_int64, c_int, c_int, c_float, P]),
    "yt8m_wimg_unregister": (c_int64, [P, P]),
    "yt8m_wimg_lookup": (c_void_p, [P, c_int64, c_int64, c_int64, c_int, c_int, c_float]),
    "yt8m_wimg_count": (c_int64, []),
    "yt8m_wimg_watch": (c_int, [P, P, c_int]),
    "yt8m_wimg_note_demand": (c_int, [P, c_int64, c_int64, c_int64, c_int, c_int, c_float]),
    "yt8m_wimg_demands": (c_int64, [ctypes.POINTER(WimgDemand), c_int64]),
    "yt8m_wimg_demand_generation": (c_int64, [P, P]),
    "yt8m_wimg_jobs_layout": (c_int64, [ctypes.POINTER(WimgJob), c_int64]),
    "yt8m_adam_tiles": (c_int, [P, P, P, P, P, c_int64, c_int64, c_int64, P, c_float, P, c_float, c_float, c_float, c_float, c_float,
                                c_int, P]),
    "yt8m_adam_multi_ex": (c_int, [P, P, P, P, P, c_int64, P, c_float, P, c_float, c_float, c_float, c_float, c_float, P, P]),
    "yt8m_optimizer_ranges": (c_int, [ctypes.POINTER(OptRanges), P]),
    "yt8m_lstm_stack_set_early_optimizer": (c_int, [ctypes.POINTER(OptRanges)]),
    "yt8m_h2_split": (c_int, [P, c_int64, c_int64, c_int64, c_float, P, P, P, P, P]),
    "yt8m_h2_absmax": (c_int, [P, c_int64, c_int64, c_int64, P, P]),
    "yt8m_h2_degraded": (c_int, [ctypes.POINTER(ctypes.c_uint64), c_int, P]),
    "yt8m_clip_by_norm_f32": (c_int, [P, P, c_int64, c_float, P, P]),
    "yt8m_lstm_persist_set_pair": (c_int, [c_int]),
    "yt8m_gru_persist_supported": (c_int, [c_int64, c_int64]),
    "yt8m_gru_persist_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "yt8m_gru_persist_fwd": (c_int, [P, P, P, c_int64, P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_int64, P, c_int64, P]),
    "yt8m_gru_persist_bwd": (c_int, [P, P, P, c_int64, P, c_int64, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, P, c_int64, P]),
    "yt8m_h2_split_dropout": (c_int, [P, c_int64, c_int64, c_float, P, P, P, c_float, ctypes.c_uint64, c_int64, P]),
    "yt8m_h2_split_ex": (c_int, [P, c_int64, c_int64, c_int64, c_float, P, P, P, P, P, P, P, P]),
    "yt8m_gemm_h1x2_nt_ex": (c_int, [c_int64, c_int64, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_float, P, P, P, c_float, c_float, P,
                                     c_int64, P]),
    "yt8m_h2_rowscales": (c_int, [P, c_int64, c_int64, c_int64, P, P, P]),
    "yt8m_h2_split_rows": (c_int, [P, c_int64, c_int64, c_int64, P, P, P]),
    "yt8m_gemm_h2_nt_ex": (c_int, [c_int64, c_int64, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_float, P, P, P, c_float, P, c_int64, P]),
    "yt8m_timepool_max_f32": (c_int, [P, c_int64, c_int64, c_int64, c_int64, P, P, c_int64, P]),
    "yt8m_timepool_shiftmax_f32": (c_int, [P, c_int64, c_int64, c_int64, c_int, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), P, P,
                                           c_int64, P]),
    "yt8m_u8_cnn_pool_dw": (c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, P, c_float, P]),
    "yt8m_u8_frames_image_f16": (c_int, [P, P, c_int64, c_int64, c_int64, c_float, P, P, P, P]),
    "yt8m_u8_frames_image_t_f16": (c_int, [P, P, c_int64, c_int64, c_int64, P, P]),
    "yt8m_gemm_h2_nt_grouped": (c_int, [c_int, ctypes.POINTER(GemmProblem), ctypes.POINTER(c_float), PP, PP, P, c_int64, P]),
    "yt8m_topk_rows": (c_int, [P, c_int64, c_int64, c_int, P, P, P]),
    "yt8m_perr_rows": (c_int, [P, P, c_int64, c_int64, P, P]),
}

# The ABI this host binds (include/yt8m_hip.h, yt8m_abi_version): workspace layouts and argument meanings, not just symbols.
ABI_VERSION = 4

_lib = None


class Yt8mHipError(RuntimeError):
    pass


def lib():
    """Returns the loaded CDLL; raises (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Yt8mHipError(
                "libyt8m_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C youtube-8m_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        # torch first: its bundled HIP runtime must be the one the process initialises.  Loaded the other way round (build() then smoke()
        # in one process on the GPU box) the library's launches fail with "no ROCm-capable device is detected".
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.yt8m_abi_version() != ABI_VERSION:
            raise Yt8mHipError("libyt8m_hip.so speaks ABI %d, this host was written for ABI %d: rebuild with `make -C youtube-8m_amd/csrc`"
                               % (L.yt8m_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


_STATUS = {-1: "YT8M_E_BADARG", -2: "YT8M_E_SHAPE", -3: "YT8M_E_HIP", -4: "YT8M_E_RCCL"}


def check(status):
    if status != 0:
        msg = lib().yt8m_last_error().decode("utf-8", "replace")
        exc = ValueError if status in (-1, -2) else Yt8mHipError
        raise exc("%s: %s" % (_STATUS.get(status, status), msg))
