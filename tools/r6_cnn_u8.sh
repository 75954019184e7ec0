cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_models.py -x -q -m gpu -k "cnn" 2>&1 | tail -25
for m in cnn_chain; do YT8M_NO_PROF=1 python tools/model_bench.py $m 2>&1 | grep -v amdgpu | tail -3 | cut -c1-200; done
