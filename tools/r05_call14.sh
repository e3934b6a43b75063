#!/bin/bash
# Round 5, GPU call 14: the same variant at configs[4]'s own size (65 536 frames: below the threshold -- only the branch is new).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05n
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for lib in "" lpcnt; do
    for ex in "" "--lpc-exact" "--fused"; do
      env ${lib:+ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so} timeout 300 python bench.py --workload lpc $ex --no-cpu-baseline --steps 400 --warmup 20 > $O/l.json 2> /dev/null
      echo "lpc 65536 frames [${lib:-ship}] [${ex:-default}]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-110)"
    done
  done
done
