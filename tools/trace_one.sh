# usage: bash tools/trace_one.sh <model_bench config> [skip fraction] [list min_us]   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -o t -- python $GRAFT_REPO_ROOT/tools/model_bench.py $1 > /tmp/tr1.log 2>&1
grep "B=" /tmp/tr1.log | cut -c1-90
f=$(find /tmp/tr1 -name "*kernel_trace.csv")
python $GRAFT_REPO_ROOT/tools/trace_gaps.py $f ${2:-0.6}
if [ -n "$3" ]; then python $GRAFT_REPO_ROOT/tools/trace_list.py $f $3; fi
