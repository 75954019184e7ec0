"""Cross-stream hand-off cost, repeated on the same stream pair (first use of a stream pays queue creation), default and high priority."""
import os, sys, time
import torch
dev = torch.device("cuda:0")
x = torch.randn(128, 4096, device=dev)
big = torch.randn(64 << 20, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 200
def chain(s1, s2, same):
    torch.cuda.synchronize()
    for _ in range(24): big.mul_(1.0000001)
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1): x.add_(1.0); a = s1.record_event(torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    for i in range(N // 2):
        with torch.cuda.stream(s1): x.add_(1.0)
        if not same: s2.wait_stream(s1)
        with torch.cuda.stream(s1 if same else s2): x.add_(1.0)
        if not same: s1.wait_stream(s2)
    t1 = time.perf_counter()
    with torch.cuda.stream(s1): b = s1.record_event(torch.cuda.Event(enable_timing=True))
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / N, (t1 - t0) / N * 1e6
for name, mk in (("default priority", lambda: torch.cuda.Stream()), ("high priority", lambda: torch.cuda.Stream(priority=-1))):
    s1, s2 = mk(), mk()
    for rep in range(4):
        g, h = chain(s1, s2, False)
        print("%-17s rep %d: cross-stream GPU %.1f us / launch (host %.1f)" % (name, rep, g, h), flush=True)
    g, h = chain(s1, s2, True)
    print("%-17s same stream   GPU %.1f us / launch (host %.1f)" % (name, g, h), flush=True)
