# counter passes over tools/b1_bench.py on one shape set, for the one-plane GEMM kernels (b1 / phased; YT8M_B1_ALIAS=2 = no DMA in the loop)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/b1pmc; export B1_ONLY="${B1_ONLY:-sq8k}"
for p in 0 1; do for a in 0; do
  YT8M_B1_PIPE=$p rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/b1pmc/v${p}${a} -o q --output-format csv -- python tools/b1_bench.py > gpurun_out/b1pmc/v${p}${a}.log 2>&1
done; done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/b1pmc/v*/q_counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        if "gemm_b1" in k:
            c={x: sum(v)/len(v) for x,v in agg[k].items()}
            cyc=c["GRBM_GUI_ACTIVE"]/8
            print(f.split('/')[-2], k, "cycles/launch %.0f  mfma busy %.3f  wait_inst %.3f  wait_any %.3f  active_inst %.3f (of wave cycles)" % (cyc, c["SQ_VALU_MFMA_BUSY_CYCLES"]/(cyc*1024), c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"], c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"]/c["SQ_WAVE_CYCLES"]))
PY
