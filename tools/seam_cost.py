"""How much does a dependent tiny launch cost on this box without a tracer?  (decides whether merging the step's small launches pays)"""
import os, sys, time, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd._lib as L
from yt8m_amd.ops import _p, _stream
dev = torch.device("cuda:0")
lib = L.lib()
x = torch.randn(128, 4096, device=dev)
w = torch.zeros(64, dtype=torch.int32, device=dev)
big = torch.randn(64 << 20, device=dev)
N = 300
def run(name, fn, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-46s host enqueue %.1f us/launch, total %.1f us/launch" % (name, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
# a long kernel first so that the host runs ahead of the GPU: then the GPU-side seam is what is measured
def ahead(fn):
    def g():
        fn()
    return g
run("torch x.add_(1) [2 MB]", lambda: x.add_(1.0))
run("lib h2_absmax [128x4096]", lambda: lib.yt8m_h2_absmax(_p(x), 128, 4096, 4096, _p(w), _stream()))
run("hipMemsetAsync via torch zero_ [256 B]", lambda: w.zero_())
# GPU-side seam with the host far ahead: queue a 5 ms kernel, then N tiny ones, time the tiny ones with events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (("torch add_", lambda: x.add_(1.0)), ("lib h2_absmax", lambda: lib.yt8m_h2_absmax(_p(x), 128, 4096, 4096, _p(w), _stream())),
                 ("zero_ 256 B", lambda: w.zero_())):
    torch.cuda.synchronize()
    for _ in range(8): big.mul_(1.0000001)          # ~ms of queued work
    e0.record()
    for _ in range(N): fn()
    e1.record()
    torch.cuda.synchronize()
    print("GPU-side, host ahead: %-20s %.2f us per dependent launch" % (name, e0.elapsed_time(e1) * 1e3 / N), flush=True)
# two streams ping-pong through events (the stack's cross-stream seams)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for _ in range(8): big.mul_(1.0000001)
e0.record()
s1.wait_stream(torch.cuda.current_stream())
for i in range(N // 2):
    with torch.cuda.stream(s1): x.add_(1.0)
    s2.wait_stream(s1)
    with torch.cuda.stream(s2): x.add_(1.0)
    s1.wait_stream(s2)
torch.cuda.current_stream().wait_stream(s1)
e1.record()
torch.cuda.synchronize()
print("GPU-side, cross-stream ping-pong: %.2f us per dependent launch" % (e0.elapsed_time(e1) * 1e3 / N), flush=True)
# the same chain, host time separated; and a one-way hand-off chain (A on s1, then B on s2 waits once per pair: the stack's pattern)
for mode in ("pingpong", "same_stream_with_events", "prio"):
    if mode == "prio":
        s1, s2 = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)
    torch.cuda.synchronize()
    for _ in range(16): big.mul_(1.0000001)
    e0.record()
    s1.wait_stream(torch.cuda.current_stream())
    s2.wait_stream(torch.cuda.current_stream())
    t0 = time.perf_counter()
    for i in range(N // 2):
        if mode == "same_stream_with_events":
            with torch.cuda.stream(s1): x.add_(1.0)
            ev = s1.record_event()
            s1.wait_event(ev)
            with torch.cuda.stream(s1): x.add_(1.0)
            ev = s1.record_event()
            s1.wait_event(ev)
        else:
            with torch.cuda.stream(s1): x.add_(1.0)
            s2.wait_stream(s1)
            with torch.cuda.stream(s2): x.add_(1.0)
            s1.wait_stream(s2)
    t1 = time.perf_counter()
    torch.cuda.current_stream().wait_stream(s1)
    torch.cuda.current_stream().wait_stream(s2)
    e1.record()
    torch.cuda.synchronize()
    print("%-26s GPU %.2f us per dependent launch (host enqueue %.2f us per launch)" % (mode, e0.elapsed_time(e1) * 1e3 / N, (t1 - t0) / N * 1e6), flush=True)
