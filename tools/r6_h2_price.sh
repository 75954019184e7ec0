cd $GRAFT_REPO_ROOT
for e in "YT8M_GEMM_H2_PRICE=0" "A=1" "YT8M_GEMM_H2_PRICE=0" "A=1"; do echo "== $e"; for m in chain config5 netvlad dbof cnn_chain lstm_attn gru_pool; do env YT8M_NO_PROF=1 $e python tools/model_bench.py $m 2>&1 | grep "ms/step" | cut -c1-60; done; done
for e in "YT8M_GEMM_H2_PRICE=0" "A=1" "YT8M_GEMM_H2_PRICE=0" "A=1"; do env $e python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gap 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$e headline: %.3f ms/step' % d['ms_per_step'], [ (e['workload'][:20], e['dtype'], e['per_gpu_batch'], round(e['ms_per_step'],3)) for e in d['extra'][:4]])"; done
