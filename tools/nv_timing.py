import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.seq_ops as seq_ops
dev = torch.device("cuda:0")
F, D, K = 300, 1152, 64
for B in (1024,):
    gen = torch.Generator(device=dev).manual_seed(1)
    q = torch.randint(0, 256, (B, F, D), device=dev, generator=gen, dtype=torch.uint8)
    nf = torch.full((B,), F, device=dev, dtype=torch.int32)
    Wc = torch.randn((D, K), device=dev, generator=gen) / D ** 0.5
    bc = torch.zeros(K, device=dev)
    for nsplit in (2, 1):
        for _ in range(3):
            cT, n, agg = seq_ops.netvlad_fwd_u8(q, nf, Wc, bc, nsplit=nsplit)
        torch.cuda.synchronize()
        print("B", B, "nsplit", nsplit, "loop cycles mean %.0f min %.0f max %.0f | epilogue mean %.0f min %.0f max %.0f"
              % (n[:, 0].mean(), n[:, 0].min(), n[:, 0].max(), n[:, 1].mean(), n[:, 1].min(), n[:, 1].max()))
        # single-pass kernel (vlad_video_kernel): phase 1 / its epilogue / phase 2 / store tail (100 MHz s_memtime ticks -> see the ratio)
        print("   video kernel sections: " + " | ".join("%s mean %.0f max %.0f" % (k, n[:, i].mean(), n[:, i].max())
                                                         for i, k in enumerate(("phase1", "epilogue1", "phase2", "tail"))))
