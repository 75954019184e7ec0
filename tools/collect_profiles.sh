#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/prof_r1/ (copied to profiles/ afterwards).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r1
mkdir -p $O
cd /tmp
# 1. headline bench under kernel trace + stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline < /dev/null > $O/bench_line.json 2> $O/bench.err
# 2. NetVLAD plugin step (config 3) kernel trace
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/netvlad -o nv -- python $R/tools/model_bench.py netvlad < /dev/null > $O/netvlad_step.txt 2>&1
# 3. fused pooling micro-bench: kernel trace, then PMC passes (separate runs)
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/nvb -o nvb -- python $R/tools/netvlad_bench.py 128 1024 < /dev/null > $O/netvlad_bench.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/nvpmc -o fetch -- python $R/tools/netvlad_bench.py 1024 < /dev/null > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/nvpmc -o write -- python $R/tools/netvlad_bench.py 1024 < /dev/null > /dev/null 2>&1
# 4. LSTM step timeline
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/lstm -o lstm -- python $R/tools/model_bench.py lstm < /dev/null > $O/lstm_step.txt 2>&1
# 5. all plugin configs, un-profiled
timeout 400 python $R/tools/model_bench.py < /dev/null > $O/model_bench.txt 2>&1
timeout 200 python $R/tools/model_bench.py config5_bf16 netvlad_bf16 lstm_bf16 config5_b1024 config5_bf16_b1024 < /dev/null >> $O/model_bench.txt 2>&1
YT8M_NO_PROF=1 timeout 100 python $R/tools/model_bench.py lstm < /dev/null 2>&1 | grep "B=" | sed 's/^lstm /lstm(no library profiler, hipGraph replay on) /' >> $O/model_bench.txt
timeout 100 python $R/tools/gemm_shapes.py < /dev/null > $O/gemm_shapes.txt 2>&1
timeout 100 python $R/tools/skinny_bench.py < /dev/null > $O/skinny_bench.txt 2>&1
timeout 100 python $R/bench.py --dtype bf16 --no-cpu-baseline < /dev/null > $O/bench_bf16_line.json 2>/dev/null
find $O -name "*.csv" | head -30
# keep the merge small: drop raw traces larger than 8 MB
find $O -name "*.csv" -size +8M -delete
