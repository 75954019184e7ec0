cd $GRAFT_REPO_ROOT
for e in "A=0" "YT8M_WIMG_H2=0" "YT8M_WIMG=0"; do echo "== $e"; for m in config5 chain netvlad dbof cnn_chain moe; do env YT8M_NO_PROF=1 $e python tools/model_bench.py $m 2>&1 | grep "ms/step" | cut -c1-60; done; done
