cd $GRAFT_REPO_ROOT
echo "== U=8 (default)"; python tools/gru_persist_check.py 300 2>&1 | tail -2
echo "== U=16 whole chip"; YT8M_GRU_BWD_U=16 python tools/gru_persist_check.py 300 2>&1 | tail -1
echo "== U=16 128 CUs";   YT8M_GRU_BWD_U=16 YT8M_GRU_BWD_CUS=128 python tools/gru_persist_check.py 300 2>&1 | tail -1
