#!/bin/bash
# Repeats the headline bench in fresh processes and keeps the per-step intervals of every run (diagnostic for the rare slow run).
export YT8M_BENCH_STEP_TIMES=${YT8M_BENCH_STEP_TIMES:-1}
N=${1:-12}
for i in $(seq 1 $N); do
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra --no-roofline 2> gpurun_out/hunt_$i.err > gpurun_out/hunt_$i.json
  python - "$i" <<'PY'
import json, sys
i = sys.argv[1]
d = json.loads(open("gpurun_out/hunt_%s.json" % i).read().strip().splitlines()[-1])
print("run %s: %.2f ms/step" % (i, d["ms_per_step"]), d.get("placement"))
if d["ms_per_step"] > 26.5:
    print(open("gpurun_out/hunt_%s.err" % i).read()[-6000:])
PY
done
