#!/bin/bash
# Second schedule sweep of round 5 (after the f16 backward recurrence and the in-recurrence maxima): tail sub-parts, part ratios, K-part slots.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-78s %.2f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@"; }
run A=warm
run A=base
run YT8M_STACK_SUB0_LAST=2
run YT8M_STACK_SUB0_LAST=3
run YT8M_STACK_BWD_PARTS=3,3,2,2
run YT8M_STACK_BWD_PARTS=2,2,1,1,1
run YT8M_STACK_BWD_PARTS=3,2,2,1
run YT8M_STACK_BWD_PARTS=4,4,3,2,1
run YT8M_STACK_BWD_PARTS=2,2,2,1
run YT8M_X3_SLOTS=128
run YT8M_X3_SLOTS=192
run YT8M_STACK_CHAIN_COMBINE=1
run A=base
