cd $GRAFT_REPO_ROOT
for m in chain netvlad cnn_chain dbof config5 lstm_attn chain_dropout lstm_mem_dropout; do YT8M_NO_PROF=1 python tools/model_bench.py $m 2>&1 | grep "ms/step" | cut -c1-45; done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gap 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('headline: %.3f ms/step' % d['ms_per_step']); [print(e['workload'][:22], e['dtype'], e['per_gpu_batch'], e['ms_per_step']) for e in d['extra']]"
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|assert |Error" | head -8
