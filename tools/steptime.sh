cd /tmp && export TMPDIR=/tmp
for v in base kdiv2 kdiv8; do
  if [ $v = base ]; then unset YT8M_LIB; else export YT8M_LIB=$GRAFT_REPO_ROOT/tools/variants/lib_$v.so; fi
  rm -rf /tmp/st_$v
  YT8M_STEP_MODE=2020 YT8M_SET=lstm_pipeline_chunks=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/st_$v -o t -- python $GRAFT_REPO_ROOT/tools/model_bench.py lstm > /tmp/st_$v.log 2>&1
  echo "== $v"; grep "B=" /tmp/st_$v.log | cut -c1-70
  python $GRAFT_REPO_ROOT/tools/trace_gaps.py $(find /tmp/st_$v -name "*kernel_trace.csv") 0.6 | grep -i "lstm_step\|lstm_gates\|window"
done
