cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|assert " | head -8
for m in netvlad moe config5 chain; do timeout 300 python tools/model_bench.py $m 2>&1 | grep "ms/step" | cut -c1-80; done
