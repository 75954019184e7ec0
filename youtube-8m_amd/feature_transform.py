"""Input transformers (W/feature_transform.py + W/all_feature_transform/default_transformer.py:4-8)."""
import torch

from . import ops
from .flags import DEFINE_string

DEFINE_string("feature_transformer", "DefaultTransformer", "how to preprocess feature, defaults to identical")


class DefaultTransformer(object):
    """L2-normalise the feature axis.  uint8 inputs are the raw reader bytes: dequantise (W/utils.py:23-38),
    zero the padding rows (W/readers.py:178-187) and normalise in ONE pass over the uint8 block."""

    def transform(self, model_input_raw, num_frames, **unused_params):
        if model_input_raw.dtype == torch.uint8:
            return ops.dequant_l2norm(model_input_raw, num_frames), num_frames
        return ops.l2norm_fwd(model_input_raw), num_frames


class IdenticalTransformer(object):
    def transform(self, model_input_raw, num_frames, **unused_params):
        return model_input_raw, num_frames
