"""One shape of the b1 kernel a few times (PMC passes: tools/pmc_run.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd._lib as L
import yt8m_amd.ops as ops
from yt8m_amd.ops import _p, _stream
dev = torch.device("cuda:0")
lib = L.lib()
M, N, K = 8192, 14148, 2304
A32, B32 = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
ia, ib = ops.bf16_image(A32), ops.bf16_image(B32)
out = torch.empty((M, N), device=dev)
pr = (L.GemmProblem * 1)(L.GemmProblem(M, N, K, ia.buf.data_ptr(), 0, ib.buf.data_ptr(), 0, out.data_ptr(), N, None, 0.0))
ws = ops._workspace(dev)
for _ in range(4):
    L.check(lib.yt8m_gemm_b1_nt_grouped(1, pr, _p(ws), ws.numel() * 4, _stream()))
torch.cuda.synchronize()
