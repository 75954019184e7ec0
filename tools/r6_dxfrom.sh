cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "chain or golden or multitask or cnn" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for e in "YT8M_MOE_DX_FROM=0" "A=1" "YT8M_MOE_DX_FROM=0" "A=1"; do echo "== $e"; for m in chain chain_dropout cnn_chain config5; do env YT8M_NO_PROF=1 $e python tools/model_bench.py $m 2>&1 | grep "ms/step" | cut -c1-75; done; done
