#!/bin/bash
# A/B of the image-GEMM main-loop schedules (round-3 kernels vs the interleaved round-4 kernels; YT8M_B1_PIPE / YT8M_X3_PIPE) on the
# GPU box: unit shapes (tools/b1_bench.py, tools/x3_time.py), the matrix-pipe counters of both one-plane kernels at 8192^3 (own
# counter pass, kernel trace only), the zero-operand probe (clock headroom), and the two end-to-end steps that use them.
# Output: gpurun_out/gemm_ab/summary.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gemm_ab
rm -rf $O; mkdir -p $O
cd /tmp
{
for p in 0 1; do
  echo "== one-plane bf16 GEMM, YT8M_B1_PIPE=$p (0: gemm_b1_kernel, 1: gemm_b1q_kernel), N(0,1) operands"
  YT8M_B1_PIPE=$p timeout 200 python $R/tools/b1_bench.py 2>/dev/null | grep "image b1" | cut -c1-12,42-90
  echo "== the same, zero-filled operands (B1_ZEROS=1: no operand toggling -> the clock the power limit takes away)"
  for sh in sq8k "head fwd"; do B1_ZEROS=1 B1_ONLY="$sh" YT8M_B1_PIPE=$p timeout 100 python $R/tools/b1_bench.py 2>/dev/null | grep "image b1" | cut -c1-12,42-90; done
done
for p in 0 1; do
  echo "== fp32-from-bf16 GEMM (six products), YT8M_X3_PIPE=$p (0: gemm_x3_kernel, 1: gemm_x3q_kernel)"
  YT8M_X3_PIPE=$p timeout 100 python $R/tools/x3_time.py 2>/dev/null
done
echo "== counters at 8192^3 (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); cycles per launch = GRBM_GUI_ACTIVE / 8)"
for p in 0 1; do
  B1_ONLY=sq8k YT8M_B1_PIPE=$p rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_b1_$p -o q -- python $R/tools/b1_bench.py > /dev/null 2>&1
  python $R/tools/pmc_mfma_busy.py $O/pmc_b1_$p/q_counter_collection.csv | grep "gemm_b1"
  YT8M_X3_PIPE=$p rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_x3_$p -o q -- python $R/tools/x3_time.py > /dev/null 2>&1
  python $R/tools/pmc_mfma_busy.py $O/pmc_x3_$p/q_counter_collection.csv | grep "gemm_x3"
done
echo "== end to end (ms/step): configs[4] bf16 B = 1024 and the configs[3] bf16 variant by YT8M_B1_PIPE; the headline by YT8M_X3_PIPE"
for p in 0 1 0 1; do
  echo "YT8M_B1_PIPE=$p $(YT8M_B1_PIPE=$p timeout 250 python $R/tools/model_bench.py config5_bf16_b1024 lstm_bf16 2>/dev/null | grep ms/step | cut -c1-52 | tr '\n' '|')"
done
for p in 0 1 0 1; do
  echo "YT8M_X3_PIPE=$p headline $(YT8M_X3_PIPE=$p timeout 250 python $R/bench.py --no-extra --no-cpu-baseline --no-gap --steps 30 2>/dev/null | tail -1 | python -c 'import json,sys; print(round(json.load(sys.stdin)["ms_per_step"],2))') ms/step"
done
} > $O/summary.txt 2>&1
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt
