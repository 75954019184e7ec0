#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python tools/persist_check.py time 2>&1 | grep -E "OK|FAIL|bwd kernel" | cut -c1-150
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_x3.py tests/test_gpu_round2.py -m gpu -q -p no:cacheprovider -k "lstm or Lstm or recurrence or persist or stack or headline" 2>&1 | grep -E "passed|failed" | tail -1
for i in 1 2; do timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra --no-roofline 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],2))"; done
