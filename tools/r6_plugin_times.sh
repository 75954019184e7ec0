cd $GRAFT_REPO_ROOT
L="logistic moe chain lstm lstm_attn netvlad config5 lstm_parallel lstm_posattn cnn_chain gru_pool ln_lstm lstm_mem_dropout chain_dropout dbof"
for m in $L; do python tools/model_bench.py $m 2>&1 | grep "ms/step"; done > gpurun_out/r6_plugin_step_times.txt
echo "(per-family hipEvent profiler on: hipGraph replay off.  The same call with YT8M_NO_PROF=1:" >> gpurun_out/r6_plugin_step_times.txt
for m in $L; do YT8M_NO_PROF=1 python tools/model_bench.py $m 2>&1 | grep "ms/step" | cut -c1-62; done >> gpurun_out/r6_plugin_step_times.txt
cat gpurun_out/r6_plugin_step_times.txt | cut -c1-110
