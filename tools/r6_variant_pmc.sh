#!/bin/bash
# Round 6: stand-alone recurrence timing + the two decisive counter groups for a library variant.  usage: r6_variant_pmc.sh <variant name> [more...]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for v in "$@"; do
  O=$R/gpurun_out/pmc_$v
  rm -rf $O; mkdir -p $O
  lib=$R/tools/variants/lib_$v.so
  [ "$v" = "base" ] && lib=$R/youtube-8m_amd/libyt8m_hip.so
  YT8M_LIB=$lib python $R/tools/pmc_recur.py 3 > $O/unprofiled.txt 2>&1
  YT8M_LIB=$lib timeout 150 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/g1 -o g1 -- python $R/tools/pmc_recur.py 3 < /dev/null > $O/g1.log 2>&1
  YT8M_LIB=$lib timeout 150 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/g2 -o g2 -- python $R/tools/pmc_recur.py 3 < /dev/null > $O/g2.log 2>&1
  YT8M_LIB=$lib timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $O/g3 -o g3 -- python $R/tools/pmc_recur.py 3 < /dev/null > $O/g3.log 2>&1
  python $R/tools/pmc_recur_summary.py $O lstm_persist_bwd > $O/summary.txt 2>&1
  find $O -name "*.csv" -size +4M -delete
  echo "===== variant $v"; cat $O/summary.txt
done
