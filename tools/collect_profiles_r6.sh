#!/bin/bash
# Collects the round-6 rocprofv3 evidence on the GPU box into gpurun_out/prof_r6/ (tools/make_profile_docs_r6.py copies the
# summaries to profiles/ afterwards).  PMC passes are their own runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r6
rm -rf $O; mkdir -p $O
cd /tmp
# 1. PMC passes of the headline step FIRST: the bench line of step 2 cites the per-launch traffic they give (profiles/r6_pmc_traffic_lstm.json)
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc -o fetch -- python $R/tools/pmc_run_lstm.py 5 < /dev/null > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc -o write -- python $R/tools/pmc_run_lstm.py 5 < /dev/null > /dev/null 2>&1
# 1b. matrix-pipe utilisation of the dominant kernels (north_star: "MFMA-busy against chip peak"): one more counter pass, kernel
#     trace only.  SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs) per kernel -> tools/pmc_mfma_busy.py
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o mfma -- python $R/tools/pmc_run_lstm.py 5 < /dev/null > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc -o sq -- python $R/tools/pmc_run_lstm.py 5 < /dev/null > /dev/null 2>&1
python $R/tools/pmc_mfma_busy.py $(find $O/pmc -name "mfma*counter_collection.csv" | head -1) > $O/mfma_busy.txt 2>&1
(cd $R && python tools/make_profile_docs_r6.py pmc > $O/pmc_traffic.log 2>&1; cp profiles/r6_pmc_traffic_lstm.json $O/ 2>/dev/null)
# 2. the headline bench line exactly as the driver runs it (un-profiled, all legs incl. the >= 10-step CPU baseline)
timeout 900 python $R/bench.py --steps 20 --warmup 5 < /dev/null > $O/bench_line.json 2> $O/bench.err
cp $R/bench_extra.json $O/bench_extra.json          # the sidecar the line names (full per-family detail of every configuration)
# 3. the same workload under kernel trace + stats (no CPU / GAP / extra legs: they are not the measured region)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra < /dev/null > $O/bench_line_profiled.json 2> $O/bench_prof.err
f=$(find $O/bench -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_step.py $f 40 10 u8_frames_tm_kernel > $O/step_timeline.txt 2>&1   # (a step of the TIMED region: the last five steps of the run carry the hipEvent profile pass)
# 4. extras under kernel trace: configs[1], configs[2], configs[4] (bf16)
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/moe -o moe -- python $R/bench.py --workload moe --steps 200 --warmup 10 --no-cpu-baseline --no-gap --no-extra < /dev/null > $O/moe_line.json 2>/dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/netvlad -o nv -- python $R/bench.py --workload netvlad --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra < /dev/null > $O/netvlad_line.json 2>/dev/null
# 4b. the single-pass NetVLAD forward (opt-in) next to the rows + cols pair: step time, per-kernel time, HBM-side bytes
YT8M_NO_PROF=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o c5 -- python $R/tools/model_bench.py config5_bf16_b1024 < /dev/null > $O/config5_bf16.txt 2>&1
# 5. un-profiled micro-benches
timeout 200 python $R/tools/persist_check.py time < /dev/null > $O/persist_check.txt 2>&1
timeout 100 python $R/tools/fwd_pair_check.py < /dev/null > $O/fwd_pair_check.txt 2>&1
timeout 200 python $R/tools/x3_check.py time < /dev/null > $O/x3_check.txt 2>&1
timeout 100 python $R/tools/gemm_shapes.py lstm < /dev/null > $O/gemm_shapes_lstm.txt 2>&1
timeout 200 python $R/tools/b1_bench.py < /dev/null > $O/b1_bench.txt 2>&1
timeout 600 python $R/tools/model_bench.py < /dev/null > $O/model_bench.txt 2>&1
timeout 100 python $R/tools/gru_persist_check.py 300 < /dev/null > $O/gru_persist_check.txt 2>&1
timeout 400 python $R/tools/reader_bench.py --threads 4,8,16 < /dev/null > $O/reader_bench.txt 2>&1
# keep the merge small: drop raw traces larger than 6 MB (the per-kernel stats CSVs stay)
find $O -name "*.csv" -size +6M -delete
find $O -name "*.csv" | head -40
