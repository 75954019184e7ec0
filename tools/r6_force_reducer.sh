#!/bin/bash
# Round 6: the data-parallel machinery on ONE MI355X with the CU footprint of an 8-GPU all-reduce emulated
# (YT8M_DP_EMULATE = cus:world:busbw -> a kernel holding `cus` CUs for the ring all-reduce time of each bucket, behind the real 1-rank
# RCCL collective).  Output: one line per run (tools/make_profile_docs_r6.py turns them into profiles/r6_force_reducer.md).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline $FLAGS 2>/dev/null > /tmp/fr_line.json
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open("/tmp/fr_line.json").read())
x = {}
try:
    x = json.load(open(d.get("sidecar", "bench_extra.json")))
except Exception:
    pass
tr = None
for r in ((x.get("reducer") or {}).get("per_rank") or []):
    tr = r.get("dp_trace") or tr
out = {"run": sys.argv[1], "ms_per_step": round(d["ms_per_step"], 3), "reducer": d.get("reducer") or x.get("reducer")}
if tr:
    out["backward_ms"] = round(tr["backward_ms"], 3)
    out["exposed_allreduce_ms"] = round(tr["exposed_allreduce_ms"], 3)
    out["buckets"] = [[round(b["MB"], 1), round(b["enqueued_ms"], 2), round(b["landed_ms"], 2)] for b in tr["buckets"]]
print(json.dumps(out))
PY
}
FLAGS="" run plain A=1
FLAGS="--force-reducer" run reducer_world1 A=1
FLAGS="--force-reducer" run emu8_300_reserve0 YT8M_DP_EMULATE=32:8:300 YT8M_DP_RESERVED_CUS=0
FLAGS="--force-reducer" run emu8_300_reserve32 YT8M_DP_EMULATE=32:8:300 YT8M_DP_RESERVED_CUS=32
FLAGS="--force-reducer" run emu8_300_auto YT8M_DP_EMULATE=32:8:300
FLAGS="--force-reducer" run emu8_100_reserve0 YT8M_DP_EMULATE=32:8:100 YT8M_DP_RESERVED_CUS=0
FLAGS="--force-reducer" run emu8_100_reserve32 YT8M_DP_EMULATE=32:8:100 YT8M_DP_RESERVED_CUS=32
FLAGS="--force-reducer" run emu8_50_reserve0 YT8M_DP_EMULATE=32:8:50 YT8M_DP_RESERVED_CUS=0
FLAGS="--force-reducer" run emu8_50_auto YT8M_DP_EMULATE=32:8:50 YT8M_DP_BUSBW_GBPS=50
FLAGS="--force-reducer" run emu8_300_64cus_reserve0 YT8M_DP_EMULATE=64:8:300 YT8M_DP_RESERVED_CUS=0
FLAGS="" run plain_again A=1
