"""Reads bench.py's JSON line on stdin and prints the headline + family times (tuning aid)."""
import json
import sys

line = [l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
r = d.get("roofline") or {}
fam = {k: round(v["ms_per_step"], 2) for k, v in (r.get("families") or {}).items()}
oth = {k: round(v["ms_per_step"], 2) for k, v in (r.get("other_families") or {}).items()}
print("%s %.2f ms/step %.0f videos/s  families(ms, profiled serially) %s %s" % (tag, d["ms_per_step"], d["value"], fam, oth))
if d.get("placement"):
    print("   placement:", d["placement"])
for e in d.get("extra") or []:
    print("   extra: %s  %.3f ms/step  %.0f videos/s" % (e.get("workload", "")[:70], e.get("ms_per_step", 0), e.get("value", 0)))
