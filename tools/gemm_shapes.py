"""TFLOP/s of the fp32 grouped GEMM on a list of shapes: python tools/gemm_shapes.py  (GPU box)"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.ops as ops

dev = torch.device("cuda:0")
SHAPES = [  # (label, transA, transB, [(M, N, K), ...])
    ("cfg1 fwd  M=1024", False, False, [(1024, 14148, 1152), (1024, 9432, 1152)]),
    ("head fwd  M=8192", False, False, [(8192, 14148, 1152), (8192, 9432, 1152)]),
    ("head fwd  M=8192 one", False, False, [(8192, 23580, 1152)]),
    ("head fwd  M=8192 N=23552", False, False, [(8192, 23552, 1152)]),
    ("head fwd  M=4096", False, False, [(4096, 14148, 1152), (4096, 9432, 1152)]),
    ("head fwd  M=2048", False, False, [(2048, 14148, 1152), (2048, 9432, 1152)]),
    ("lstm proj M=38400", False, False, [(38400, 4096, 1152)]),
    ("square 4096", False, False, [(4096, 4096, 4096)]),
    ("head dW   K=8192", True, False, [(1152, 14148, 8192), (1152, 9432, 8192)]),
    ("head dx   K=23580", False, True, [(8192, 1152, 14148), (8192, 1152, 9432)]),
]
if len(sys.argv) > 1 and sys.argv[1] == "lstm":       # the hoisted products of BASELINE configs[3] (B = 128, F = 300)
    SHAPES = [
        ("L0 proj  [38400,1152]x[1152,4096]", False, False, [(38400, 4096, 1152)]),
        ("L1 proj  [38400,1024]x[1024,4096]", False, False, [(38400, 4096, 1024)]),
        ("L1 dx    [38400,4096]x[1024,4096]^T", False, True, [(38400, 1024, 4096)]),
        ("L0 dWx   [38400,1152]^T x dz", True, False, [(1152, 4096, 38400)]),
        ("L dWh    [38400,1024]^T x dz", True, False, [(1024, 4096, 38400)]),
        ("L0 dWx+dWh grouped", True, False, [(1152, 4096, 38400), (1024, 4096, 38400)]),
        ("chunk dWx+dWh K=9600 grouped", True, False, [(1152, 4096, 9600), (1024, 4096, 9600)]),
        ("chunk proj M=9600", False, False, [(9600, 4096, 1152)]),
        ("chunk dx M=9600", False, True, [(9600, 1024, 4096)]),
        ("head fwd [128,4096]x[4096,23580]", False, False, [(128, 14148, 4096), (128, 9432, 4096)]),
        ("head dW  [128,4096]^T x dZ", True, False, [(4096, 14148, 128), (4096, 9432, 128)]),
        ("head dx", False, True, [(128, 4096, 14148), (128, 4096, 9432)]),
    ]
_w = torch.randn(4096, 4096, device=dev)
for _ in range(60):                      # ~70 ms of load first: the clocks of an idle box take a while to ramp
    ops.gemm(_w, _w)
torch.cuda.synchronize()
for label, tA, tB, probs in SHAPES:
    items, fl = [], 0.0
    for (M, N, K) in probs:
        A = torch.randn((K, M) if tA else (M, K), device=dev)
        B = torch.randn((N, K) if tB else (K, N), device=dev)
        items.append(dict(A=A, B=B, out=torch.empty((M, N), device=dev)))
        fl += 2.0 * M * N * K
    for _ in range(3):
        ops.gemm_grouped(items, transA=tA, transB=tB)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        ops.gemm_grouped(items, transA=tA, transB=tB)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%-26s %8.3f ms  %6.1f TFLOP/s" % (label, ms, fl / ms / 1e9))
    del items
