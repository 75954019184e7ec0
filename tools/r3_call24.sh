#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
run() { env "$@" timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %.2f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@"; }
run A=warm
run A=base
for p in 2,3,1 2,2,1,1 1,2,2,1 3,3,2,1 5,4,2,1 4,4,3,1 3,2,1 8,6,4,2,1 2,2,2; do run YT8M_STACK_BWD_PARTS=$p; done
run A=base
