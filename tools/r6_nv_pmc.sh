#!/bin/bash
# Round 6, VERDICT r5 #2: TCP / TCC counter passes of the NetVLAD forward kernels (rows + cols pair, and the single pass), to price them
# against the CU's vector-memory window found in profiles/r6_pmc_recur_tcc.txt.  One group per run, kernel trace only.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_nv
rm -rf $O; mkdir -p $O
cd /tmp
for mode in 0 1; do
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  YT8M_NETVLAD_SINGLE=$mode timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/m${mode}g$i -o g$i -- python $R/tools/nv_pmc_run.py 4 < /dev/null > $O/m${mode}g$i.log 2>&1
  echo "m${mode}g$i rc=$? : $grp" >> $O/groups.txt
done <<'EOG'
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_BUSY_sum
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS
EOG
done
python $R/tools/pmc_recur_summary.py $O vlad_ > $O/summary.txt 2>&1
for mode in 0 1; do YT8M_NETVLAD_SINGLE=$mode timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$mode -o t -- python $R/tools/nv_pmc_run.py 20 > /dev/null 2>&1; grep vlad_ $(find $O/t$mode -name "*kernel_stats.csv" | head -1) | cut -c1-60,150-260 >> $O/summary.txt; done
find $O -name "*.csv" -size +4M -delete
cat $O/summary.txt
# LN-LSTM kernels after the load hoisting
YT8M_NO_PROF=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ln -o ln -- python $R/tools/model_bench.py ln_lstm > $O/ln.txt 2>&1
grep "ms/step" $O/ln.txt; head -8 $(find $O/ln -name "*kernel_stats.csv" | head -1) | cut -c1-100,160-240
