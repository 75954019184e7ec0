"""Helpers mirroring W/utils.py that sit on the hot path."""
import logging

import torch

from . import ops


def Dequantize(feat_vector, max_quantized_value=2, min_quantized_value=-2):
    """W/utils.py:23-38 (affine map only; the fused device path is ops.dequant_l2norm)."""
    assert max_quantized_value > min_quantized_value
    quantized_range = max_quantized_value - min_quantized_value
    scalar = quantized_range / 255.0
    bias = (quantized_range / 512.0) + min_quantized_value
    return feat_vector.to(torch.float32) * scalar + bias


def GetListOfFeatureNamesAndSizes(feature_names, feature_sizes):
    """W/utils.py:140-161."""
    list_of_feature_names = [n.strip() for n in feature_names.split(",")]
    list_of_feature_sizes = [int(s) for s in feature_sizes.split(",")]
    if len(list_of_feature_names) != len(list_of_feature_sizes):          # the reference logs and carries on (:155-158)
        logging.error("length of the feature names (=" + str(len(list_of_feature_names)) + ") != length of feature "
                      "sizes (=" + str(len(list_of_feature_sizes)) + ")")
    return list_of_feature_names, list_of_feature_sizes


def clip_gradient_norms(gradients_to_variables, max_norm):
    """W/utils.py:164-174, same signature: [(grad, var)] -> [(clipped grad, var)], per tensor
    tf.clip_by_norm(g, max_norm) = g * max_norm / max(||g||, max_norm); None gradients pass through.
    API-parity form for callers that hold their own gradient list; the training step itself runs the same rule fused with
    Adam over the gradient arena (ops.sqnorm_and_adam -> yt8m_sqnorm_multi + yt8m_adam_multi), see gradient_norms()."""
    clipped_grads_and_vars = []
    for grad, var in gradients_to_variables:
        if grad is not None and grad.is_cuda:                     # device tensors: the library's kernels (yt8m_clip_by_norm_f32)
            import ctypes
            from . import _lib
            g = grad.to(torch.float32).contiguous()
            out, ws = torch.empty_like(g), torch.empty(256, dtype=torch.float32, device=g.device)
            _lib.check(_lib.lib().yt8m_clip_by_norm_f32(ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(out.data_ptr()), g.numel(),
                                                        float(max_norm), ctypes.c_void_p(ws.data_ptr()),
                                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
            grad = out
        elif grad is not None:                                    # host tensors (lists a caller assembled on the CPU): plain arithmetic
            norm = torch.linalg.vector_norm(grad.to(torch.float32))
            grad = grad * (max_norm / torch.clamp(norm, min=max_norm))
        clipped_grads_and_vars.append((grad, var))
    return clipped_grads_and_vars


def gradient_norms(graph):
    """Per-tensor gradient norms of the last fused clip+Adam pass (what W/utils.py:164-174 clips against)."""
    return graph.norms.sqrt()
