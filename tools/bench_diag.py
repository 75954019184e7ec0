"""Host-side diagnostics of the headline training step that used to live inside bench.py's timed_run ("move the diagnostics
out of the function that defines the metric"): per-step device intervals, host enqueue time per step, allocator counters, and -- with
--profile -- a cProfile of the enqueueing thread plus a 1 kHz sample of every other thread's stack and of the autograd workers' kernel state.
Not a measurement of record: it perturbs what it looks at.   usage: python tools/bench_diag.py [--steps 20] [--warmup 5] [--profile]"""
import argparse
import collections
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="lstm")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = bench.WORKLOADS[a.workload]
    g, tg, pool = bench.build(a.workload, cfg["batch"], 1, 0, dev, None, False)
    marks, host, allocs = [], [], []

    def run(k, base, record):
        for i in range(k):
            x, y, nf = pool[(base + i) % len(pool)]
            h0 = time.perf_counter()
            tg.step(x, y, nf)
            if record:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append(e)
                host.append((time.perf_counter() - h0) * 1e3)
                ms = torch.cuda.memory_stats()
                allocs.append((ms.get("num_device_alloc", -1), ms["reserved_bytes.all.current"] >> 20))

    run(max(a.warmup, 1), 0, False)
    torch.cuda.synchronize()
    prof = sampler = None
    hist, stop, me = collections.Counter(), threading.Event(), threading.get_ident()
    if a.profile:
        import cProfile
        prof = cProfile.Profile()

        def sample():
            while not stop.is_set():
                for tid, fr in sys._current_frames().items():
                    if tid == threading.get_ident():
                        continue
                    chain, f = [], fr
                    while f is not None and len(chain) < 3:
                        chain.append("%s:%d %s" % (os.path.basename(f.f_code.co_filename), f.f_lineno, f.f_code.co_name))
                        f = f.f_back
                    hist[("main " if tid == me else "other ") + " <- ".join(chain)] += 1
                for t in os.listdir("/proc/self/task"):
                    try:
                        comm = open("/proc/self/task/%s/comm" % t).read().strip()
                        if not comm.startswith("pt_autograd"):
                            continue
                        st = open("/proc/self/task/%s/stat" % t).read().rsplit(")", 1)[1].split()[0]
                        wch = open("/proc/self/task/%s/wchan" % t).read().strip()
                        hist["task %s state %s wchan %s" % (comm, st, wch)] += 1
                    except Exception:
                        pass
                time.sleep(0.001)

        sampler = threading.Thread(target=sample, daemon=True)
        sampler.start()
        prof.enable()
    t0 = time.perf_counter()
    run(a.steps, a.warmup, True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if prof is not None:
        prof.disable()
        stop.set()
        sampler.join()
        for k, v in hist.most_common(25):
            print("%6d  %s" % (v, k))
        import pstats
        pstats.Stats(prof, stream=sys.stdout).sort_stats("tottime").print_stats(18)
    print("%.3f ms/step over %d steps (diagnostic run: events recorded per step)" % (el / a.steps * 1e3, a.steps))
    print("step intervals (ms): " + " ".join("%.1f" % marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)))
    print("host enqueue per step (ms): " + " ".join("%.1f" % h for h in host))
    print("device allocs / reserved MiB after each step: " + " ".join("%d/%d" % x for x in allocs))
    ms = torch.cuda.memory_stats()
    print("allocator: free / total GB %s, reserved %.1f GB, retries %d, ooms %d" % (
        " / ".join("%.1f" % (v / 2 ** 30) for v in torch.cuda.mem_get_info()), ms["reserved_bytes.all.current"] / 2 ** 30,
        ms["num_alloc_retries"], ms["num_ooms"]))


if __name__ == "__main__":
    main()
