cd $GRAFT_REPO_ROOT
for r in 1024 512 256; do echo "== MOE_LOGITS_H2_MIN_ROWS $r"; for m in chain chain_dropout cnn_chain; do YT8M_NO_PROF=1 YT8M_MOE_LOGITS_H2_MIN_ROWS=$r python tools/model_bench.py $m 2>&1 | grep "ms/step" | cut -c1-75; done; done
echo "== repeat 1024 / 512"; for r in 1024 512; do YT8M_NO_PROF=1 YT8M_MOE_LOGITS_H2_MIN_ROWS=$r python tools/model_bench.py chain 2>&1 | grep "ms/step" | cut -c1-75; done
