"""Frame sampling / pooling helpers (mirrors W/model_utils.py:23-95) on the device (csrc/dbof.hip): the index bookkeeping is
integer work driven by one Philox uniform per (video, sample) or per video; the gathered frames are data (no gradient)."""
from . import ops
from .variables import get_default_graph


def _seed(seed):
    return get_default_graph().next_random_seed() if seed is None else int(seed)


def SampleRandomSequence(model_input, num_frames, num_samples, seed=None, return_index=False):
    """W/model_utils.py:23-48: a random contiguous run of num_samples frames (clamped to the last valid frame)."""
    return ops.sample_frames(model_input, num_frames, num_samples, 1, _seed(seed), return_index)


def SampleRandomFrames(model_input, num_frames, num_samples, seed=None, return_index=False):
    """W/model_utils.py:51-70: num_samples frames drawn uniformly (with replacement) from the valid ones."""
    return ops.sample_frames(model_input, num_frames, num_samples, 0, _seed(seed), return_index)


def FramePooling(frames, method, **unused_params):
    """W/model_utils.py:72-95."""
    if method in ("average", "max"):
        return ops.frame_pool(frames, method)
    elif method == "none":
        return frames.reshape(-1, frames.shape[2])
    else:
        raise ValueError("Unrecognized pooling method: %s" % method)
