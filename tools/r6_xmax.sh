cd $GRAFT_REPO_ROOT
for w in moe netvlad; do for i in 1 2; do python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-gap --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w %.4f ms/step' % d['ms_per_step'])"; done; done
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|assert |Error" | head -8
