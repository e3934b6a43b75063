#!/bin/bash
# the bench's own channel-major launch (separate allocations for x and y) with the staggered start off / on, same box, fresh processes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_cmrep2; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
for rep in 1 2 3; do
  for st in 0 20 80; do
    for m in "" "--fused"; do
      ALZ_DUO_STAGGER=$st timeout 300 python bench.py --workload biquad --layout chan $m --no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json - > $O/l.json 2> $O/l.err || tail -3 $O/l.err
      echo "rep $rep stagger $st [$m]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
    done
  done
done 2>&1 | tee $O/cm_repeat2.log
