"""Helpers mirroring W/utils.py that sit on the hot path."""
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
    if len(list_of_feature_names) != len(list_of_feature_sizes):
        raise ValueError("length of the feature names (=%r) != length of feature sizes (=%r)"
                         % (len(list_of_feature_names), len(list_of_feature_sizes)))
    return list_of_feature_names, list_of_feature_sizes


def clip_gradient_norms(graph, max_norm):
    """W/utils.py:164-174 is realised inside the fused optimiser pass (ops.sqnorm_and_adam): per-tensor
    g * clip / max(||g||, clip).  This helper only reports the per-tensor norms of the last step."""
    return graph.norms.sqrt()
