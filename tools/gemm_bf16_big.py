"""bf16 NT GEMM: 128x128-tile persistent kernel vs the 256x256-tile kernel on the cfg[4] head shapes; checks values too."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.ops as ops

dev = torch.device("cuda:0")


def run(items, big):
    os.environ["YT8M_BF16_BIG_MIN"] = "1" if big else "1000000000"
    for _ in range(3):
        ops.gemm_bf16_nt_grouped(items)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        outs = ops.gemm_bf16_nt_grouped(items)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, outs


w = torch.randn(4096, 4096, device=dev)
for _ in range(40):
    ops.gemm(w, w)
SHAPES = [("head fwd", [(8192, 14148, 2304), (8192, 9432, 2304)]), ("head dW", [(2304, 14148, 8192), (2304, 9432, 8192)]),
          ("head dx g", [(8192, 2304, 14148)]), ("head dx e", [(8192, 2304, 9432)]), ("cfg1 fwd", [(1024, 14148, 1152), (1024, 9432, 1152)]),
          ("B128 head fwd", [(1024, 14148, 2304), (1024, 9432, 2304)]), ("ragged", [(1000, 777, 1000), (300, 5000, 72)])]
for label, probs in SHAPES:
    items, fl = [], 0.0
    for (M, N, K) in probs:
        A = ops._bf16_empty(M, K, dev)                     # 16-byte aligned row pitch, as ops.cast_bf16 produces
        A.copy_(torch.randn(M, K, device=dev))
        B = ops._bf16_empty(N, K, dev)
        B.copy_(torch.randn(N, K, device=dev))
        items.append(dict(A=A, B=B, bias=torch.randn(N, device=dev)))
        fl += 2.0 * M * N * K
    t0, o0 = run(items, False)
    t1, o1 = run(items, True)
    err = max(float((a - b).abs().max()) / max(1.0, float(a.abs().max())) for a, b in zip(o0, o1))
    print("%-14s small-tile %7.3f ms %7.1f TF/s | big-tile %7.3f ms %7.1f TF/s | max rel diff %.2e" % (label, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, err))
