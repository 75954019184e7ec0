cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_models.py -x -q -m gpu -k "parallel" 2>&1 | tail -15
for m in lstm_parallel; do YT8M_NO_PROF=1 python tools/model_bench.py $m 2>&1 | grep "ms/step" | cut -c1-80; done
