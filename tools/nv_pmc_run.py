"""A few NetVLAD forward calls at B = 1024 (rows + cols pair, or the single pass with YT8M_NETVLAD_SINGLE=1) for rocprofv3 --pmc passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.seq_ops as seq_ops
dev = torch.device("cuda:0")
B, F, D, K = 1024, 300, 1152, 64
gen = torch.Generator(device=dev).manual_seed(1)
q = torch.randint(0, 256, (B, F, D), device=dev, generator=gen, dtype=torch.uint8)
nf = torch.full((B,), F, device=dev, dtype=torch.int32)
Wc = torch.randn((D, K), device=dev, generator=gen) / D ** 0.5
bc = torch.zeros(K, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    seq_ops.netvlad_fwd_u8(q, nf, Wc, bc, nsplit=2)
torch.cuda.synchronize()
