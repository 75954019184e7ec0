"""seq_ops.gru_layer (two stacked layers, plugin shape) with the persistent recurrences against the per-step kernels: outputs and gradients."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.seq_ops as seq_ops
from yt8m_amd.variables import reset_default_graph, xavier_uniform, zeros, ones
dev = torch.device("cuda:0")
F, B, D, H = int(os.environ.get("F", 300)), 128, 1152, 1024

def run(fwd, bwd):
    seq_ops.GRU_PERSIST_FWD, seq_ops.GRU_PERSIST_BWD = fwd, bwd
    g = reset_default_graph(device=dev, seed=0)
    g.begin_step()
    gen = torch.Generator(device=dev).manual_seed(3)
    x = (torch.rand((F, B, D), device=dev, generator=gen) * 4 - 2).requires_grad_(True)
    nf = torch.full((B,), F, device=dev, dtype=torch.int32)
    vs, d_in = [], D
    for l in range(2):
        vs.append((g.get_variable("l%d/wg" % l, (d_in + H, 2 * H), xavier_uniform), g.get_variable("l%d/bg" % l, (2 * H,), ones),
                   g.get_variable("l%d/wc" % l, (d_in + H, H), xavier_uniform), g.get_variable("l%d/bc" % l, (H,), zeros)))
        d_in = H
    g.finalize()
    h = x
    outs = []
    for l in range(2):
        h, hf = seq_ops.gru_layer(h, *vs[l], nf)
        outs.append(h)
    w = torch.randn((F, B, H), device=dev, generator=gen) * 0.01
    (h * w).sum().backward()
    torch.cuda.synchronize()
    res = {"out0": outs[0].detach().clone(), "out1": outs[1].detach().clone(), "dx": x.grad.clone()}
    for l in range(2):
        for n, v in zip(("wg", "bg", "wc", "bc"), vs[l]):
            res["l%d/%s" % (l, n)] = v.grad.clone()
    return res

a = run(False, False)
for name, fb in (("fwd only", (True, False)), ("bwd only", (False, True)), ("both", (True, True))):
    b = run(*fb)
    print(name + ": " + "  ".join("%s %.1e/%.1e%s" % (k, float((a[k] - b[k]).abs().nan_to_num(1e9).max()), float(a[k].abs().max()),
                                                      "" if bool(torch.isfinite(b[k]).all()) else " NONFINITE(%d)" % int((~torch.isfinite(b[k])).sum())) for k in a))
