#!/bin/bash
# A/B of the single-pass NetVLAD forward (vlad_video_kernel) against the rows + cols pair on BASELINE configs[2], B = 1024:
# step time un-profiled, then per-kernel durations under rocprofv3 --kernel-trace --stats, then the HBM-side byte counters of the
# forward kernels in their own --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in separate runs, kernel trace only).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_nv
rm -rf $O; mkdir -p $O
cd /tmp
for m in 1 0; do
  YT8M_NETVLAD_SINGLE=$m timeout 200 python $R/bench.py --workload netvlad --steps 40 --warmup 5 --no-cpu-baseline --no-gap --no-extra < /dev/null > $O/line_single$m.json 2> $O/err_single$m.txt
  grep "timed region" $O/err_single$m.txt | sed "s/^/single=$m /"
done
for m in 1 0; do
  YT8M_NETVLAD_SINGLE=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$m -o nv -- python $R/bench.py --workload netvlad --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra --no-roofline < /dev/null > /dev/null 2> $O/prof$m.err
  f=$(find $O/prof$m -name "*kernel_stats.csv" | head -1)
  echo "== single=$m kernel stats"; head -14 $f | cut -c1-160
  cp $f $O/kernel_stats_single$m.csv
done
if [ "$1" = "pmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    for m in 1 0; do
      YT8M_NETVLAD_SINGLE=$m timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_$m -o nv -- python $R/bench.py --workload netvlad --steps 4 --warmup 2 --no-cpu-baseline --no-gap --no-extra --no-roofline < /dev/null > /dev/null 2> $O/pmc_${c}_$m.err
      f=$(find $O/pmc_${c}_$m -name "*counter_collection.csv" | head -1)
      python $R/tools/pmc_kernels.py $f vlad_ > $O/pmc_${c}_single$m.txt 2>&1
      echo "== $c single=$m"; cat $O/pmc_${c}_single$m.txt
    done
  done
fi
find $O -name "*.csv" -size +3M -delete
find $O -type d -name "prof*" -prune -o -type f -print | head -30
