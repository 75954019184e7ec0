"""LayerNorm-LSTM layer 0 at full size: byte path (U8FrameImages) against the float path on the same weights: z (hoisted projection),
outputs, states; and the weight gradient of a random functional."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.seq_ops as seq_ops
import yt8m_amd.ops as ops
from yt8m_amd.variables import reset_default_graph, xavier_uniform, ones, zeros
from oracle import np_ref
dev = torch.device("cuda:0")
B, F, D, H = int(os.environ.get("B", 128)), int(os.environ.get("F", 300)), 1152, 1024
gen = torch.Generator(device=dev).manual_seed(1)
q = torch.randint(0, 256, (B, F, D), device=dev, generator=gen, dtype=torch.uint8)
nf = torch.full((B,), F, device=dev, dtype=torch.int32)
res = {}
for mode in ("float", "u8"):
    g = reset_default_graph(device=dev, seed=0)
    W = g.get_variable("w", (D + H, 4 * H), xavier_uniform)
    gam = [g.get_variable("g%d" % k, (H,), ones) for k in range(5)]
    bet = [g.get_variable("b%d" % k, (H,), zeros) for k in range(5)]
    g.finalize(); g.begin_step()
    if mode == "u8":
        x = seq_ops.U8FrameImages(q, nf)
        # the projection alone
        z = torch.empty((F * B, 4 * H), device=dev)
        seq_ops.u8_hoisted_fwd(x, W.data[:D], None, z)
    else:
        x = ops.dequant_l2norm(q, nf).transpose(0, 1).contiguous()
        z = ops.gemm_any(x.view(F * B, D), W.data[:D], role=ops._hoisted_role(F * B, 4 * H, D, False))
    out, c, h = seq_ops.lnlstm_layer(x, W, gam, bet, nf)
    coef = torch.randn(out.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    (out * coef).sum().backward()
    res[mode] = (z.double().cpu(), out.detach().double().cpu(), c.detach().double().cpu(), W.grad[:D].double().cpu(), W.grad[D:].double().cpu())
x64 = torch.from_numpy(np_ref.dequant_l2norm_folded(q.cpu().numpy(), nf.cpu().numpy())).transpose(0, 1).reshape(F * B, D)
g = reset_default_graph(device=dev, seed=0)
W = g.get_variable("w", (D + H, 4 * H), xavier_uniform)
zr = x64 @ W.data[:D].double().cpu()
for k, name in enumerate(("z", "out", "c", "dW_x", "dW_h")):
    a, b = res["float"][k], res["u8"][k]
    print("%-5s float vs u8: max diff %.3g (scale %.3g)" % (name, float((a - b).abs().max()), float(a.abs().max())), flush=True)
print("z vs fp64: float path %.3g, u8 path %.3g (scale %.3g)" % (float((res["float"][0] - zr).abs().max()), float((res["u8"][0] - zr).abs().max()), float(zr.abs().max())))
per_t = (res["float"][1] - res["u8"][1]).abs().amax(dim=(1, 2))
print("out diff per step: first 5 %s ... last 5 %s" % (per_t[:5].tolist(), per_t[-5:].tolist()))
