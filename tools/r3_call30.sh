#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
run() { env "$@" timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra --no-roofline 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-60s %.2f' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@"; }
run A=plain; run GPU_MAX_HW_QUEUES=2; run GPU_MAX_HW_QUEUES=3; run GPU_MAX_HW_QUEUES=5; run GPU_MAX_HW_QUEUES=16; run A=plain
