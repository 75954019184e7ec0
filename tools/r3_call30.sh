#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
run() { env "$@" timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra --no-roofline $EXTRA 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-60s %.2f' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@" $EXTRA; }
EXTRA=""; run A=plain; run GPU_MAX_HW_QUEUES=8
EXTRA="--force-reducer"; run YT8M_DP_LAYER_BUCKETS=0; run YT8M_DP_LAYER_BUCKETS=0 GPU_MAX_HW_QUEUES=8; run YT8M_DP_LAYER_BUCKETS=1; run YT8M_DP_LAYER_BUCKETS=1 GPU_MAX_HW_QUEUES=8; run YT8M_DP_LAYER_BUCKETS=1 GPU_MAX_HW_QUEUES=6
