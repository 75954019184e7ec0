#!/bin/bash
# Round-5 schedule knobs of the native LSTM stack re-measured with the h2 products in place (the weight-gradient stream is ~2x less
# loaded than in round 4, the dx chain is four short launches): one line per configuration, headline workload, 20 timed steps.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-78s %.2f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@"; }
run A=warm
run A=base
run YT8M_STACK_DX_STREAM=1
run YT8M_STACK_BWD_PARTS=1,2,2,1
run YT8M_STACK_DX_STREAM=1 YT8M_STACK_BWD_PARTS=1,2,2,1
run YT8M_STACK_DX_STREAM=1 YT8M_STACK_BWD_PARTS=1,1,2,1,1
run YT8M_STACK_DX_STREAM=1 YT8M_STACK_BWD_PARTS=1,1,1,1,1,1
run YT8M_STACK_DX_STREAM=1 YT8M_STACK_BWD_PARTS=1,1,1,1
run YT8M_STACK_DX_STREAM=1 YT8M_STACK_BWD_PARTS=1,2,2,2,2,2,1
run YT8M_STACK_DX_STREAM=1 YT8M_STACK_SUB0_LAST=2
run YT8M_STACK_DX_STREAM=1 YT8M_STACK_SW2=1
run YT8M_STACK_DX_STREAM=1 YT8M_STACK_SW2=1 YT8M_STACK_BWD_PARTS=1,2,2,1
run YT8M_STACK_BWD_PARTS=1,1,1,1,1,1
run YT8M_STACK_SW2=1
run A=base
