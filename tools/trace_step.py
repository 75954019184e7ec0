"""One training step of a rocprofv3 kernel trace as a timeline: kernels >= min_us in start order with start offset, duration
and stream (queue).  The step is found between two consecutive launches of a kernel that runs once per step (default adam_chunk_kernel; the LSTM step runs the optimiser twice since round 4: pass u8_frames_tm_kernel).
usage: python tools/trace_step.py <kernel_trace.csv> [min_us] [which step from the end, default 2] [delimiter kernel]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
mn = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
delim = sys.argv[4] if len(sys.argv) > 4 else "adam_chunk_kernel"     # a kernel launched exactly once per step
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
adam = [i for i, e in enumerate(ev) if delim in e[2]]
lo, hi = adam[-back - 1], adam[-back]
t0 = ev[lo][1]
print("step: %.3f ms between optimizer launches" % ((ev[hi][1] - ev[lo][1]) / 1e6))
queues = {}
for s, e, n, q in ev[lo + 1:hi + 1]:
    queues.setdefault(q, len(queues))
    if (e - s) / 1e3 >= mn:
        k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
        print("%8.3f -> %8.3f ms  %8.1f us  q%-2d %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, queues[q], k))
