#!/bin/bash
# Round 6 (last session): backward partition of the native stack re-swept on the final recurrences (register-carried epilogue, f16 forms)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" timeout 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %.3f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@"; }
run A=warm
run A=base
run YT8M_STACK_BWD_PARTS=1,2,2,1
run YT8M_STACK_BWD_PARTS=1,2,2,1,1
run YT8M_STACK_BWD_PARTS=4,4,2,1,1
run YT8M_STACK_BWD_PARTS=3,3,2,2
run YT8M_STACK_BWD_PARTS=3,2,2,1
run YT8M_STACK_BWD_PARTS=2,2,2,1,1
run YT8M_STACK_BWD_PARTS=3,3,2,1,1
run YT8M_STACK_BWD_PARTS=2,1,1
run YT8M_STACK_BWD_PARTS=1,1,1,1
run YT8M_STACK_BWD_PARTS=5,5,3,2
run A=base
run YT8M_STACK_SUB0_LAST=2
run YT8M_EARLY_ADAM=0
run YT8M_LSTM_PERSIST_FWD_CHUNKS=2
run A=base
