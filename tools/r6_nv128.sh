cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "netvlad or vlad or c2_ or config5 or c4_" 2>&1 | grep -E "passed|failed|Error|assert |timed out|error" | head -10
timeout 300 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "gru" 2>&1 | grep -E "passed|failed" | head -3
for k in 1 0 1 0; do echo "== K128=$k"; YT8M_NETVLAD_K128=$k python tools/netvlad_bench.py 1024 2>&1 | grep "nsplit"; done
for k in 1 0; do echo "== K128=$k bench netvlad"; YT8M_NETVLAD_K128=$k python bench.py --workload netvlad --steps 30 --warmup 5 --no-cpu-baseline --no-gap --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); h=d['roofline'].get('hbm',{})
print('%.3f ms/step' % d['ms_per_step'], {k:(round(v.get('avg_launch_us',0),1) if isinstance(v,dict) else v) for k,v in h.items() if isinstance(v,dict)}, h.get('forward_8d'))"; done
cd /tmp; export TMPDIR=/tmp
for k in 1 0; do YT8M_NETVLAD_K128=$k timeout 100 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_nv128_$k -o g -- python $GRAFT_REPO_ROOT/tools/nv_pmc_run.py 4 > /dev/null 2>&1; echo "== K128=$k counters"; python $GRAFT_REPO_ROOT/tools/pmc_recur_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_nv128_$k vlad_rows | head -5; done
