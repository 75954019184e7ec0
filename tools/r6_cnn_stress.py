"""Shape sweep of seq_ops.u8_cnn_maxpool / u8_cnn against fp64 (edge shapes: F below the filter length, one video group, wide D, column
counts on both pooling forms, empty videos)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.seq_ops as seq_ops
from yt8m_amd.variables import reset_default_graph, zeros
from oracle import np_ref
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
worst = 0.0
for (B, F, D, shapes) in [(16, 1, 16, [(1, 32), (2, 32), (3, 64)]), (16, 2, 64, [(1, 8), (2, 8), (3, 12)]), (48, 5, 1152, [(1, 128), (2, 128), (3, 256)]),
                          (128, 3, 128, [(3, 32)]), (32, 40, 96, [(1, 64), (4, 32)]), (16, 300, 32, [(1, 4), (2, 4), (3, 8)])]:
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = rs.randint(0, F + 1, size=B).astype(np.int32)
    nf[0] = F
    qd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(nf).to(dev)
    assert seq_ops.u8_cnn_supported(qd), (B, F, D)
    Ws = [(rs.randn(fs * D, n) * 0.1).astype(np.float32) for fs, n in shapes]
    coef = rs.randn(B, sum(n for _, n in shapes)).astype(np.float32)
    g = reset_default_graph(device=dev, seed=0)
    fv = [g.get_variable("f%d" % k, W.shape, zeros) for k, W in enumerate(Ws)]
    g.finalize()
    for v, W in zip(fv, Ws):
        v.data.copy_(torch.from_numpy(W).to(dev))
    g.begin_step()
    frames = seq_ops.U8FrameImages(qd, nfd)
    p = seq_ops.u8_cnn_maxpool(frames, fv)
    (p * torch.from_numpy(coef).to(dev)).sum().backward()
    x = torch.from_numpy(np_ref.dequant_l2norm_folded(q, nf))
    tw = [torch.from_numpy(W.astype(np.float64)).requires_grad_(True) for W in Ws]
    cols = []
    for (fs, n), W in zip(shapes, tw):
        sh = [x] + [torch.cat([x.new_zeros(B, min(i, F), D), x[:, :max(F - i, 0)]], dim=1) for i in range(1, fs)]
        cols.append(torch.cat(sh, dim=2) @ W)
    pr = torch.cat(cols, dim=2).max(dim=1).values
    (pr * torch.from_numpy(coef.astype(np.float64))).sum().backward()
    e1 = float(np.abs(p.detach().cpu().numpy() - pr.detach().numpy()).max() / max(1.0, float(pr.detach().abs().max())))
    e2 = max(float(np.abs(v.grad.cpu().numpy() - t.grad.numpy()).max() / max(1.0, float(t.grad.abs().max()))) for v, t in zip(fv, tw))
    worst = max(worst, e1, e2)
    print("B %3d F %3d D %4d %-32s pooled err %.2e grad err %.2e" % (B, F, D, shapes, e1, e2), flush=True)
assert worst < 1e-5, worst
print("ok, worst", worst)
