#!/bin/bash
# Forward layer wavefront of the native stack: forward chunks (YT8M_LSTM_PERSIST_FWD_CHUNKS) with whole-chip forward launches
# (default: 256 workgroups, two launches cannot run side by side -- tools/fwd_pair_check.py) and with half-chip ones (YT8M_PERSIST_CUS=128).
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-78s %.2f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@"; }
run A=warm
run A=base
run YT8M_LSTM_PERSIST_FWD_CHUNKS=2
run YT8M_LSTM_PERSIST_FWD_CHUNKS=3
run YT8M_LSTM_PERSIST_FWD_CHUNKS=4
run YT8M_PERSIST_CUS=128
run YT8M_PERSIST_CUS=128 YT8M_LSTM_PERSIST_FWD_CHUNKS=2
run YT8M_PERSIST_CUS=128 YT8M_LSTM_PERSIST_FWD_CHUNKS=3
run YT8M_PERSIST_CUS=128 YT8M_LSTM_PERSIST_FWD_CHUNKS=4
run YT8M_PERSIST_CUS=128 YT8M_LSTM_PERSIST_FWD_CHUNKS=6
run A=base
