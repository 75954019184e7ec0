cd $GRAFT_REPO_ROOT
for h in 1 0 1 0; do echo "== MOE_DX_H2=$h"; YT8M_MOE_DX_H2=$h python bench.py --workload netvlad --steps 30 --warmup 6 --no-cpu-baseline --no-gap --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.4f ms/step' % d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['roofline']['families'].items()})"; done
timeout 900 python -m pytest tests -x -q -m gpu -k "moe or chain or netvlad or fullsize" 2>&1 | grep -E "passed|failed|assert |Error" | head -5
