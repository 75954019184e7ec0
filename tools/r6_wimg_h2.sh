cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "resident_half" 2>&1 | grep -E "passed|failed|assert |Error|error" | head -8
for h in 1 0 1 0; do echo "== WIMG_H2=$h"; YT8M_WIMG_H2=$h python bench.py --workload moe --steps 200 --warmup 10 --no-cpu-baseline --no-gap --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.4f ms/step' % d['ms_per_step'])"; done
