#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/pmc_b1; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 120 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O -o hit -- python $R/tools/b1_pmc.py < /dev/null > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- python $R/tools/b1_pmc.py < /dev/null > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum --output-format csv -d $O -o req -- python $R/tools/b1_pmc.py < /dev/null > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O -o sq -- python $R/tools/b1_pmc.py < /dev/null > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete
for f in $(find $O -name "*counter_collection.csv"); do echo "== $f"; grep "gemm_b1" $f | head -12 | cut -c1-400; done
