"""Frame sampling / pooling helpers (mirrors W/model_utils.py:23-95).  Index bookkeeping is integer work done
with torch index ops on the device; the gathered frames are data (no gradient)."""
import torch


def SampleRandomSequence(model_input, num_frames, num_samples, generator=None):
    """W/model_utils.py:23-48: a random contiguous run of num_samples frames (clamped to the last valid frame)."""
    batch_size = model_input.shape[0]
    dev = model_input.device
    frame_index_offset = torch.arange(num_samples, device=dev).unsqueeze(0).expand(batch_size, -1)
    nf = num_frames.reshape(batch_size, 1)
    max_start_frame_index = torch.clamp(nf - num_samples, min=0)
    u = torch.rand((batch_size, 1), device=dev, generator=generator)
    start_frame_index = (u * (max_start_frame_index + 1).to(torch.float32)).to(torch.int32)
    frame_index = torch.minimum(start_frame_index + frame_index_offset, (nf - 1).to(torch.int32))
    return gather_frames(model_input, frame_index)


def SampleRandomFrames(model_input, num_frames, num_samples, generator=None):
    """W/model_utils.py:51-70: num_samples frames drawn uniformly (with replacement) from the valid ones."""
    batch_size = model_input.shape[0]
    u = torch.rand((batch_size, num_samples), device=model_input.device, generator=generator)
    frame_index = (u * num_frames.reshape(batch_size, 1).to(torch.float32)).to(torch.int32)
    return gather_frames(model_input, frame_index)


def gather_frames(model_input, frame_index):
    idx = frame_index.long().clamp_(min=0).unsqueeze(2).expand(-1, -1, model_input.shape[2])
    return torch.gather(model_input, 1, idx)


def FramePooling(frames, method, **unused_params):
    """W/model_utils.py:72-95."""
    if method == "average":
        return frames.mean(dim=1)
    elif method == "max":
        return frames.max(dim=1).values
    elif method == "none":
        return frames.reshape(-1, frames.shape[2])
    else:
        raise ValueError("Unrecognized pooling method: %s" % method)
