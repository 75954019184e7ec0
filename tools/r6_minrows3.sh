cd $GRAFT_REPO_ROOT
for mnk in 4e9 1e9 1e18; do echo "== MIN_MNK $mnk"; for m in chain netvlad cnn_chain dbof config5 lstm_attn; do YT8M_NO_PROF=1 YT8M_LINEAR_H2_MIN_MNK=$mnk python tools/model_bench.py $m 2>&1 | grep "ms/step" | cut -c1-45; done; done
for mnk in 4e9 1e18; do YT8M_LINEAR_H2_MIN_MNK=$mnk python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gap --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('headline MNK $mnk: %.3f ms/step' % d['ms_per_step'])"; done
