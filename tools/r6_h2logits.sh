cd $GRAFT_REPO_ROOT
for h in 1 0 1 0; do echo "== MOE_LOGITS_H2=$h"; YT8M_MOE_LOGITS_H2=$h python bench.py --workload moe --steps 200 --warmup 10 --no-cpu-baseline --no-gap --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.4f ms/step' % d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['roofline']['families'].items()})"; done
for h in 1 0; do echo "== MOE_LOGITS_H2=$h lstm headline"; YT8M_MOE_LOGITS_H2=$h python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gap --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])"; done
for h in 1 0; do echo "== MOE_LOGITS_H2=$h netvlad"; YT8M_MOE_LOGITS_H2=$h python bench.py --workload netvlad --steps 30 --warmup 5 --no-cpu-baseline --no-gap --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])"; done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|assert " | head -8
