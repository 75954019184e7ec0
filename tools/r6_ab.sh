#!/bin/bash
# Round 6 A/B: headline step time (bench.py, no extra legs) + stand-alone recurrence rates for library variants, interleaved on ONE box.
# usage: tools/r6_ab.sh <repeats> <variant> [variant ...]   ("base" = the in-tree library)
R=${GRAFT_REPO_ROOT:-/root/repo}
rep=$1; shift
cd $R
for i in $(seq 1 $rep); do
  for v in "$@"; do
    lib=$R/tools/variants/lib_$v.so; [ "$v" = "base" ] && lib=$R/youtube-8m_amd/libyt8m_hip.so
    YT8M_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gap --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); f=d['roofline']['families']; o=d['roofline'].get('other_ms_per_step',{})
print('bench %-10s %.3f ms/step  fwd-rec %.2f  bwd-rec %.2f  gemm_h2 %.2f  h1x2 %.2f  elementwise %.2f  optimizer %.2f' % ('$v', d['ms_per_step'], f['lstm_recurrence']['ms_per_step'], f['lstm_recurrence_bwd']['ms_per_step'], f['gemm_h2']['ms_per_step'], f['gemm_h1x2']['ms_per_step'], o.get('elementwise',0), o.get('optimizer',0)))"
  done
done
cd /tmp
for v in "$@"; do
  lib=$R/tools/variants/lib_$v.so; [ "$v" = "base" ] && lib=$R/youtube-8m_amd/libyt8m_hip.so
  echo "== stand-alone $v"; YT8M_LIB=$lib python $R/tools/pmc_recur.py 3 2>&1 | grep "h2:" | tail -4
done
