cd $GRAFT_REPO_ROOT
echo "== build + smoke (one process)"; python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== smoke alone"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== default bench line"; t0=$(date +%s); python bench.py > gpurun_out/r6_bench_final.json 2> gpurun_out/r6_bench_final.err; echo "wall $(( $(date +%s) - t0 )) s"; python -c "
import json; s=open('gpurun_out/r6_bench_final.json').read(); d=json.loads(s.strip().splitlines()[-1]); print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','dtype','scaling','vs_baseline')}); print('roofline', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], 'line bytes', len(s))"
