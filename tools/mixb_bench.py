"""Times yt8m_moe_mix_bwd_bf16_images at the configs[4] stage shape (B*A = 8192 rows, V = 4716, M = 2)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd._lib as L
from yt8m_amd.ops import _p, _stream
dev = torch.device("cuda:0")
lib = L.lib()
B, V, M = 8192, 4716, 2
g = torch.Generator(device=dev).manual_seed(0)
Zg = torch.randn((B, V * 3), device=dev, generator=g)
Ze = torch.randn((B, V * 2), device=dev, generator=g)
y = (torch.rand((B, V), device=dev, generator=g) < 0.001).to(torch.uint8)
kb = lambda K: (K + 15) // 16
mk = lambda rows, K: torch.empty(max(lib.yt8m_x3_image_bytes(rows, K) // 3, 16), dtype=torch.uint8, device=dev)
gi, gti, ei, eti = mk(B, V * 3), mk(V * 3, B), mk(B, V * 2), mk(V * 2, B)
part = torch.empty((lib.yt8m_moe_mix_bwd_bf16_partial_rows(B), V * 2), device=dev)
def run():
    L.check(lib.yt8m_moe_mix_bwd_bf16_images(_p(Zg), _p(Ze), None, _p(y), 0, B, V, M, 1e-6, 1.0 / B, None, _p(gi), kb(V * 3), _p(gti), kb(B),
                                             _p(ei), kb(V * 2), _p(eti), kb(B), _p(part), _stream()))
for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
rd, wr = B * V * 5 * 4 + B * V, B * V * 5 * 2 * 2
print("mix_bwd_bf16_images %.1f us  (%.2f TB/s over %.0f MB read + %.0f MB written)  checksum %d" % (
    t * 1e3, (rd + wr) / t / 1e9, rd / 1e6, wr / 1e6, int(gti.long().sum() % 1000003)))
