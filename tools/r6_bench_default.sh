cd $GRAFT_REPO_ROOT
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err
echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s  line bytes=$(wc -c < gpurun_out/r6_bench_default.json)"
tail -4 gpurun_out/r6_bench_default.err
cp bench_extra.json gpurun_out/r6_bench_extra.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6_bench_default.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('metric','value','unit','ms_per_step','n_gpus','steps','warmup','dtype')})
print('roofline', {k:d['roofline'].get(k) for k in ('kernel','achieved','peak','frac','traffic','bound')})
print('exchange', d['roofline'].get('exchange'))
print('cpu_baseline', {k:d['cpu_baseline'].get(k) for k in ('value','cores','kind','all_cores')})
PY
