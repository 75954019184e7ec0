"""A/B of the packed vs generic recurrent products of the GRU / LN-LSTM layers at full size (run twice: with and without
YT8M_NO_PACKED_CELLS=1) -- prints checksums of outputs and gradients."""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.seq_ops as seq_ops
from yt8m_amd.variables import reset_default_graph, zeros

dev = torch.device("cuda:0")
B, F, Din, H = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 20, 1152, 1024
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(F, B, Din).astype(np.float32) * 0.03).to(dev).requires_grad_(True)
nf = torch.full((B,), F, dtype=torch.int32, device=dev)
g = reset_default_graph(device=dev)
g.begin_step()
shapes = [(Din + H, 2 * H), (2 * H,), (Din + H, H), (H,)]
vs = [g.get_variable("v%d" % i, s, zeros) for i, s in enumerate(shapes)]
g.finalize()
for v in vs:
    v.data.copy_(torch.from_numpy((rs.randn(*v.shape) * 0.03).astype(np.float32)).to(dev))
out, h = seq_ops.gru_layer(x, vs[0], vs[1], vs[2], vs[3], nf)
(out.sum() * 1e-3 + (h * h).sum()).backward()
torch.cuda.synchronize()
print("out", float(out.double().abs().sum()), "h", float(h.double().abs().sum()), "nan", bool(torch.isnan(out).any()))
for v in vs:
    print("grad", tuple(v.shape), float(v.grad.double().abs().sum()))
print("dx", float(x.grad.double().abs().sum()))
