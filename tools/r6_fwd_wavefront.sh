#!/bin/bash
# Round 6: forward layer wavefront on the f16 forward recurrence -- half-chip launches (YT8M_STACK_FWD_HALF=1: 128 workgroups
# of eight tiles) with layer 1 one time chunk behind layer 0, against the serial whole-chip form.  One box, interleaved.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gap --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); f=d['roofline']['families']
print('%-34s %.3f ms/step  fwd-rec launches/step %.0f  %.2f ms  bwd-rec %.2f ms' % ('$tag', d['ms_per_step'], f['lstm_recurrence'].get('launches_per_step', 0) or 0, f['lstm_recurrence']['ms_per_step'], f['lstm_recurrence_bwd']['ms_per_step']))"
}
for rep in 1 2; do
run "serial whole chip, 1 chunk" A=1
run "whole chip, 4 chunks" YT8M_LSTM_PERSIST_FWD_CHUNKS=4
run "half chip wavefront, 2 chunks" YT8M_STACK_FWD_HALF=1 YT8M_LSTM_PERSIST_FWD_CHUNKS=2
run "half chip wavefront, 4 chunks" YT8M_STACK_FWD_HALF=1 YT8M_LSTM_PERSIST_FWD_CHUNKS=4
run "half chip wavefront, 6 chunks" YT8M_STACK_FWD_HALF=1 YT8M_LSTM_PERSIST_FWD_CHUNKS=6
run "half chip wavefront, 10 chunks" YT8M_STACK_FWD_HALF=1 YT8M_LSTM_PERSIST_FWD_CHUNKS=10
done
