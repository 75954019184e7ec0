import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.seq_ops as seq_ops
from oracle import np_ref
dev = torch.device("cuda:0")
B, F, Dm, K = 3, 300, 1152, 64
rs = np.random.RandomState(B * 1000 + F)
q = rs.randint(0, 256, size=(B, F, Dm)).astype(np.uint8)
nf = np.array([F, F, F], dtype=np.int32)
Wc = (rs.randn(Dm, K) * 3.0 / np.sqrt(Dm)).astype(np.float32)
bc = (rs.randn(K) * 0.5).astype(np.float32)
xa = np_ref.dequant_l2norm_folded(q, None)
aref = np_ref.softmax(xa @ Wc.astype(np.float64) + bc, axis=2)
for trial in range(3):
    cT, n, agg = seq_ops.netvlad_fwd_u8(torch.from_numpy(q).to(dev), torch.from_numpy(nf).to(dev), torch.from_numpy(Wc).to(dev), torch.from_numpy(bc).to(dev), nsplit=2)
    norm = np.sqrt(((q.astype(np.float64) * (4.0 / 255.0) + (4.0 / 512.0 - 2.0)) ** 2).sum(-1))
    a = cT.cpu().numpy().astype(np.float64)[:, :, :F].transpose(0, 2, 1) * norm[:, :, None]
    err = np.abs(a - aref)
    bad = np.argwhere(err > 2e-6)
    print("trial", trial, "max err", err.max(), "n bad", len(bad), "first bad", bad[:5].tolist(), "bad frames", sorted(set(bad[:, 1].tolist()))[:40])
