#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 600 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -k "vlad or netvlad or NetVLAD or config5 or composite or fixture or golden" 2>&1 | tail -8
timeout 300 python bench.py --workload netvlad --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-gap 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('netvlad B=1024', d['ms_per_step'], d['value'])"
timeout 300 python tools/model_bench.py netvlad 2>&1 | grep "B=" | cut -c1-300
YT8M_NO_PROF=1 timeout 300 python tools/model_bench.py config5_bf16_b1024 2>&1 | grep "B=" | cut -c1-100
