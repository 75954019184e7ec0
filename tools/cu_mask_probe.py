"""Where do workgroups of a CU-masked stream land?  Prints, per mask pattern, the CUs used per XCD (tuning aid for the
masked GEMM streams of the LSTM backward pass)."""
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd._lib as L  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
lib = L.lib()


def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    h = ctypes.c_void_p()
    L.check(lib.yt8m_stream_create_cu_mask(arr, len(words), ctypes.byref(h)))
    return torch.cuda.ExternalStream(h.value, device=dev)


def placement(stream, blocks=4096, spin=2000):
    out = torch.full((blocks, 2), -1, dtype=torch.int32, device=dev)
    stream.wait_stream(torch.cuda.current_stream())
    L.check(lib.yt8m_probe_placement(ctypes.c_void_p(out.data_ptr()), blocks, spin, ctypes.c_void_p(stream.cuda_stream)))
    stream.synchronize()
    o = out.cpu().numpy()
    per = collections.defaultdict(set)
    for x, hw in o:
        per[int(x)].add((int(hw) >> 8) & 0xFF)            # cu_id[11:8], sh_id[12], se_id[15:13]
    return per


def show(tag, per):
    tot = sum(len(v) for v in per.values())
    print("%-34s %3d CUs: %s" % (tag, tot, " ".join("x%d:%d" % (k, len(per[k])) for k in sorted(per))), flush=True)


show("unmasked", placement(torch.cuda.current_stream()))
pats = {
    "low 128 bits": [0xFFFFFFFF] * 4 + [0] * 4,
    "high 128 bits": [0] * 4 + [0xFFFFFFFF] * 4,
    "even bits": [0x55555555] * 8,
    "odd bits": [0xAAAAAAAA] * 8,
    "low half of each word": [0x0000FFFF] * 8,
    "bits 8..15 of every 16": [0xFF00FF00] * 8,
    "first word only": [0xFFFFFFFF] + [0] * 7,
    "one word (32 bits) all ones": [0xFFFFFFFF],
    "one word 0x0000FFFF": [0x0000FFFF],
}
for tag, words in pats.items():
    try:
        s = masked_stream(words)
        per = placement(s)
        show(tag, per)
        if tag in ("first word only",):
            print("   ", {k: sorted(v) for k, v in per.items()})
    except Exception as e:  # noqa: BLE001
        print(tag, "failed:", e)
