#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/prof_r2/ (summaries are copied to profiles/ afterwards).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r2
mkdir -p $O
cd /tmp
# 1. the headline bench line exactly as the driver runs it (un-profiled)
timeout 500 python $R/bench.py --steps 20 --warmup 3 < /dev/null > $O/bench_line.json 2> $O/bench.err
# 2. the same command under kernel trace + stats (no CPU / GAP / extra legs: they are not the measured region)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra < /dev/null > $O/bench_line_profiled.json 2> $O/bench_prof.err
# 3. PMC passes of the headline step (separate runs, kernel trace only, as the guide prescribes)
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc -o fetch -- python $R/tools/pmc_run_lstm.py 3 < /dev/null > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc -o write -- python $R/tools/pmc_run_lstm.py 3 < /dev/null > /dev/null 2>&1
# 4. extras under kernel trace: configs[1] and configs[2]
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/moe -o moe -- python $R/bench.py --workload moe --steps 200 --warmup 10 --no-cpu-baseline --no-gap --no-extra < /dev/null > $O/moe_line.json 2>/dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/netvlad -o nv -- python $R/bench.py --workload netvlad --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra < /dev/null > $O/netvlad_line.json 2>/dev/null
# 5. un-profiled micro-benches
timeout 200 python $R/tools/persist_check.py time < /dev/null > $O/persist_check.txt 2>&1
timeout 200 python $R/tools/x3_check.py time < /dev/null > $O/x3_check.txt 2>&1
timeout 100 python $R/tools/probe_mfma.py < /dev/null > $O/probe_mfma.txt 2>&1
timeout 100 python $R/tools/gemm_shapes.py lstm < /dev/null > $O/gemm_shapes_lstm.txt 2>&1
timeout 400 python $R/tools/model_bench.py < /dev/null > $O/model_bench.txt 2>&1
find $O -name "*.csv" | head -40
# keep the merge small: drop raw traces larger than 6 MB (the per-kernel stats CSVs stay)
find $O -name "*.csv" -size +6M -delete
