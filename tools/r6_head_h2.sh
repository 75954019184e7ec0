cd $GRAFT_REPO_ROOT
line() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gap --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1: %.3f ms/step' % d['ms_per_step'])"; }
line default
YT8M_MOE_LOGITS_H2_MIN_ROWS=128 line moe_h2_rows128
line default
YT8M_MOE_LOGITS_H2_MIN_ROWS=128 line moe_h2_rows128
YT8M_MOE_LOGITS_H2_MIN_ROWS=128 YT8M_MOE_DX_H2=0 line moe_h2_rows128_nodx
