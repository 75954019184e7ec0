#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python tools/model_bench.py cnn_chain netvlad chain 2>&1 | grep "B=" | cut -c1-330
