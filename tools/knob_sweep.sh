run() { env "$@" timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %.2f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@"; }
run A=warm
run A=base
run YT8M_SET=lstm_pipeline_chunks=1 YT8M_LSTM_BWD_CHUNKS=2
run YT8M_SET=lstm_pipeline_chunks=1 YT8M_LSTM_BWD_CHUNKS=3
run YT8M_SET=lstm_pipeline_chunks=1 YT8M_LSTM_BWD_CHUNKS=4
run YT8M_LSTM_BWD_CHUNKS=3
run YT8M_LSTM_BWD_CHUNKS=4
run YT8M_LSTM_BWD_CHUNKS=6
run YT8M_SET=lstm_pipeline_chunks=3
run YT8M_LSTM_FWD_WAVEFRONT=2
run YT8M_LSTM_FWD_WAVEFRONT=2 YT8M_LSTM_FWD_WAVEFRONT_CHUNKS=6
run YT8M_LSTM_FWD_WAVEFRONT=2 YT8M_LSTM_BWD_CHUNKS=4
run A=base
