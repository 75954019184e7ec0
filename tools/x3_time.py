import os, sys, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.ops as ops
dev = torch.device("cuda:0")
def timing(M, N, K, reps=10):
    A = torch.randn((M, K), device=dev); B = torch.randn((N, K), device=dev)
    ia, _ = ops.x3_split(A); ib, _ = ops.x3_split(B)
    out = torch.empty((M, N), device=dev)
    fn = lambda: ops.gemm_x3_grouped([dict(A=ia, B=ib, out=out)])
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(os.environ.get("YT8M_LIB", "default"), M, N, K, "%.3f ms  %.0f TF-eq" % (ms, 2.0 * M * N * K / ms / 1e9), flush=True)
timing(8192, 8192, 8192)
timing(19200, 4096, 1024)
