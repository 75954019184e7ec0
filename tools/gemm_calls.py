"""Per-call timing of every grouped fp32 GEMM inside one training step of a model_bench configuration (synchronising, so the
step itself is slower): python tools/gemm_calls.py config5_b1024"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.ops as ops

orig = ops.gemm_grouped
LOG = []


def timed(items, transA=False, transB=False):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(items, transA, transB)
    e1.record()
    torch.cuda.synchronize()
    fl, desc = 0.0, []
    for it in items:
        A, B = it["A"], it["B"]
        M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
        N = B.shape[0] if transB else B.shape[1]
        fl += 2.0 * M * N * K
        desc.append("%dx%dx%d ld(%d,%d) beta%g%s" % (M, N, K, A.stride(0), B.stride(0), it.get("beta", 0.0), " bias" if it.get("bias") is not None else ""))
    LOG.append((e0.elapsed_time(e1), fl, "tA%d tB%d " % (transA, transB) + " | ".join(desc)))
    return out


import model_bench as mb
name = sys.argv[1]
mb.run(name, steps=1)          # warm-up inside
ops.gemm_grouped = timed
LOG.clear()
mb.run(name, steps=1)
for ms, fl, d in LOG[-(len(LOG) // 3):]:
    if ms > float(os.environ.get("MIN_MS", "0.3")):
        print("%8.3f ms %6.1f TF/s  %s" % (ms, fl / ms / 1e9, d))
