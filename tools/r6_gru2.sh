cd $GRAFT_REPO_ROOT
for kv in "1 1" "1 0" "0 1" "0 0"; do set -- $kv; echo "== FWD=$1 BWD=$2"; YT8M_GRU_PERSIST_FWD=$1 YT8M_GRU_PERSIST_BWD=$2 timeout 300 python tools/model_bench.py gru_pool 2>&1 | grep "ms/step"; done
