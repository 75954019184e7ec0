"""Second PMC workload: copy probe (calibration) + the streaming attention-logit kernels at M = 307200, K = 1152, N = 8 + the
large-tile bf16 GEMM on one MoE-head forward shape ([8192, 2304] x [23580, 2304]^T) + the one-pass dual bf16 cast."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd._lib as L  # noqa: E402
import yt8m_amd.ops as ops  # noqa: E402

dev = torch.device("cuda:0")
lib = L.lib()
n = 64 * 1024 * 1024
a = torch.empty(n, device=dev).normal_()
b = torch.empty_like(a)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    lib.yt8m_probe_copy_f32(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), n, st)
del a, b
M, K, N = 307200, 1152, 8
x = torch.randn(M, K, device=dev)
W = torch.randn(K, N, device=dev)
dy = torch.randn(M, N, device=dev)
y = torch.empty(M, N, device=dev)
dW = torch.empty(K, N, device=dev)
dx = torch.empty(M, K, device=dev)
for _ in range(4):
    ops.skinny_fwd(x, W, None, y)
    ops.skinny_dw(x, dy, dW)
    ops.skinny_dx(dy, W, dx=dx)
del x, dx
Z = torch.randn(8192, 14148, device=dev)
for _ in range(4):
    zb, zt = ops.cast_bf16_both(Z)
A = ops._bf16_empty(8192, 2304, dev)
A.copy_(torch.randn(8192, 2304, device=dev))
Bm = ops._bf16_empty(23580, 2304, dev)
Bm.copy_(torch.randn(23580, 2304, device=dev))
out = torch.empty(8192, 23580, device=dev)
for _ in range(4):
    ops.gemm_bf16_nt_grouped([dict(A=A, B=Bm, out=out)])
torch.cuda.synchronize()
print("pmc workload 2 done")
