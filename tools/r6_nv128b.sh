cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "netvlad or vlad or c2_" 2>&1 | grep -E "passed|failed" | head -3
YT8M_NETVLAD_K128=1 timeout 900 python -m pytest tests -x -q -m gpu -k "netvlad or vlad or c2_" --deselect tests/test_gpu_round5.py::test_netvlad_single_pass_equals_the_rows_cols_pair 2>&1 | grep -E "passed|failed" | head -3
