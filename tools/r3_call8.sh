#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error" | head -20
timeout 600 python bench.py --no-cpu-baseline > $O/r3_c8_bench.json 2> $O/r3_c8_bench.err; python tools/bench_brief.py full < $O/r3_c8_bench.json; grep -i "error\|Traceback" $O/r3_c8_bench.err | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
