#!/bin/bash
# k_acorr_stage with rotating chunk register sets (shipped build) against the copied history (tools/variants/libalzhip_tuning.so built
# BEFORE the change), interleaved; then the LPC tests through the new library
cd ${GRAFT_REPO_ROOT:-$(pwd)}
R=$(pwd)
for rep in 1 2; do
  for lib in new old; do
    for m in "" "--lpc-exact" "--fused" "--lpc-frames 1048576" "--lpc-frames 1048576 --lpc-exact"; do
      if [ $lib = old ]; then export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so; else unset ALZ_LIBRARY; fi
      python bench.py --workload lpc $m --no-cpu-baseline --no-secondary --steps 200 --warmup 50 --full-json - > /tmp/l.json 2> /tmp/l.err || tail -3 /tmp/l.err
      echo "$lib [$m]: $(python tools/show_line.py /tmp/l.json | head -1 | cut -c1-120)"
    done
  done
done
unset ALZ_LIBRARY
python -m pytest tests/test_gpu_lpc.py tests/test_gpu_fullwidth.py -x -q -k "lpc or LPC or frames" 2>&1 | tail -3
