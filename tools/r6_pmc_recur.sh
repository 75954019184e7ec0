#!/bin/bash
# Round 6, VERDICT r5 #1: the TCP / TCC / UTCL1 counter passes the persistent recurrences never had.  One counter group per run
# (kernel trace only, as MI355X_MICROARCH.md prescribes).  Output: gpurun_out/pmc_recur/<group>/..., summary by tools/pmc_recur_summary.py
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_recur
rm -rf $O; mkdir -p $O
cd /tmp
python $R/tools/pmc_recur.py 3 > $O/unprofiled.txt 2>&1
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -o g$i -- python $R/tools/pmc_recur.py 3 < /dev/null > $O/g$i.log 2>&1
  echo "g$i rc=$? : $grp" >> $O/groups.txt
done <<'EOF'
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum
TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum
TCC_BUSY_avr TCC_CYCLE_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum
TCC_STREAMING_REQ_sum TCC_NC_REQ_sum TCC_UC_REQ_sum TCC_CC_REQ_sum
EOF
python $R/tools/pmc_recur_summary.py $O > $O/summary.txt 2>&1
find $O -name "*.csv" -size +4M -delete
cat $O/summary.txt
