"""Measured ceilings of this box: fp32 MFMA pipe (register-only loop) and HBM float4 copy."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd._lib as L  # noqa: E402


def measure(verbose=True):
    lib = L.lib()
    dev = torch.device("cuda:0")
    sink = torch.zeros(4, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    for blocks in (256, 512, 768, 1024):
        iters = 2000
        lib.yt8m_probe_mfma_f32(100, blocks, ctypes.c_void_p(sink.data_ptr()), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.yt8m_probe_mfma_f32(iters, blocks, ctypes.c_void_p(sink.data_ptr()), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tf = blocks * 4 * iters * 32 * 4096.0 / ms / 1e9
        out["mfma_f32_tflops_%dwg" % blocks] = tf
        if verbose:
            print("mfma f32 probe: %4d workgroups  %8.3f ms  %7.1f TFLOP/s" % (blocks, ms, tf))
    n = 256 * 1024 * 1024
    a = torch.empty(n, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(2):
        lib.yt8m_probe_copy_f32(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), n, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.yt8m_probe_copy_f32(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), n, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    out["hbm_copy_tbps"] = 2.0 * n * 4 / ms / 1e9
    if verbose:
        print("hbm copy probe: %.1f GB in %.3f ms -> %.2f TB/s (read+write)" % (2.0 * n * 4 / 1e9, ms, out["hbm_copy_tbps"]))
    return out


if __name__ == "__main__":
    measure()
