#!/bin/bash
# Round 5, GPU call 13: non-temporal tile DMA in k_acorr_stage for batches >= 512 MiB (variant library), 2^20 frames.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05m
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for lib in "" lpcnt; do
    for ex in "" "--lpc-exact"; do
      env ${lib:+ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so} timeout 300 python bench.py --workload lpc --lpc-frames 1048576 $ex --no-cpu-baseline --steps 40 --warmup 5 > $O/l.json 2> /dev/null
      echo "lpc 2^20 frames [${lib:-ship}] [${ex:-default}]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-130)"
    done
  done
done
