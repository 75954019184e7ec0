"""Times the fused uint8 NetVLAD pooling (yt8m_netvlad_fwd_u8 / bwd_u8) and prints the HBM roofline fraction.
usage: python tools/netvlad_bench.py [B ...]      (run under rocprofv3 --kernel-trace for the per-kernel split)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd.seq_ops as seq_ops  # noqa: E402

dev = torch.device("cuda:0")
F, D, K = 300, 1152, 64


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    for B in [int(a) for a in sys.argv[1:]] or [128, 1024]:
        gen = torch.Generator(device=dev).manual_seed(1)
        q = torch.randint(0, 256, (B, F, D), device=dev, generator=gen, dtype=torch.uint8)
        nf = torch.full((B,), F, device=dev, dtype=torch.int32)
        Wc = torch.randn((D, K), device=dev, generator=gen) / D ** 0.5
        bc = torch.zeros(K, device=dev)
        dagg = torch.randn((B, K, D), device=dev, generator=gen) * 1e-3
        dn = torch.randn((B, K), device=dev, generator=gen) * 1e-3
        dW, db = torch.empty((D, K), device=dev), torch.empty(K, device=dev)
        for nsplit in (2, 1):
            a, _, agg = seq_ops.netvlad_fwd_u8(q, nf, Wc, bc, nsplit=nsplit)
            tf = timeit(lambda: seq_ops.netvlad_fwd_u8(q, nf, Wc, bc, nsplit=nsplit))
            tb = timeit(lambda: seq_ops.netvlad_bwd_u8(q, nf, a, dagg, dn, dW, 0.0, db, 0.0, nsplit=nsplit))
            # algorithmic HBM bytes: fwd reads q once, writes a and agg; bwd reads q, a, dagg once
            bytes_f = B * (F * D + F * K * 4 + K * D * 4)
            bytes_b = B * (F * D + F * K * 4 + K * D * 4)
            flop = 2.0 * 2.0 * B * F * D * K
            print("B=%5d nsplit=%d  fwd %.3f ms (%.2f TB/s algorithmic, %.0f TFLOP/s)   bwd %.3f ms (%.2f TB/s, %.0f TFLOP/s)"
                  % (B, nsplit, tf, bytes_f / tf / 1e9, flop / tf / 1e9, tb, bytes_b / tb / 1e9, flop / tb / 1e9), flush=True)


if __name__ == "__main__":
    main()
