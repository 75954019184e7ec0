import os, sys, ctypes, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd._lib as L
lib = L.lib()
dev = torch.device("cuda:0")
sink = torch.zeros(4, device=dev)
for name, fn, fl in (("f32", lib.yt8m_probe_mfma_f32, 4096.0), ("bf16 const", lambda i, b, s, st: lib.yt8m_probe_mfma_bf16(i, b, 0, s, st), 32768.0),
                     ("bf16 random", lambda i, b, s, st: lib.yt8m_probe_mfma_bf16(i, b, 1, s, st), 32768.0)):
    for blocks in (512, 1024):
        for iters in (2000, 20000):
            fn(100, blocks, ctypes.c_void_p(sink.data_ptr()), None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(iters, blocks, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            print(name, "blocks", blocks, "iters", iters, "%.3f ms" % ms, "%.0f TF" % (blocks * 4 * iters * 32 * fl / ms / 1e9), flush=True)
