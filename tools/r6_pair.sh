cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "pair or carries" 2>&1 | grep -E "passed|failed|Error|assert|timed out" | head
cd /tmp
for m in 1 0; do echo "== YT8M_PERSIST_BWD_PAIR=$m"; YT8M_PERSIST_BWD_PAIR=$m timeout 200 python $GRAFT_REPO_ROOT/tools/pmc_recur.py 3 2>&1 | grep "bwd h2"; done
cd $GRAFT_REPO_ROOT
for i in 1 2; do for m in 1 0; do YT8M_PERSIST_BWD_PAIR=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gap --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); f=d['roofline']['families']
print('pair=$m  %.3f ms/step  fwd-rec %.2f  bwd-rec %.2f' % (d['ms_per_step'], f['lstm_recurrence']['ms_per_step'], f['lstm_recurrence_bwd']['ms_per_step']))"; done; done
