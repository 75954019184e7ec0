#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_x3.py tests/test_gpu_round3.py -m gpu -q --timeout=600 -p no:cacheprovider -k "persist or lstm or recurrence or native or headline" 2>&1 | tail -4
echo "== deferred arrival"; timeout 200 python tools/persist_check.py time 2>&1 | grep -v amdgpu | tail -8
echo "== eager (round-2 order)"; YT8M_LIB=$R/tools/variants/lib_eager.so timeout 200 python tools/persist_check.py time 2>&1 | grep -v amdgpu | tail -8
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline"
for v in new eager new eager; do if [ $v = eager ]; then export YT8M_LIB=$R/tools/variants/lib_eager.so; else unset YT8M_LIB; fi; timeout 300 $B 2>/dev/null | python tools/bench_brief.py $v | head -1; done
