#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for i in 1 2 3; do timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -1; done
