#!/bin/bash
# channel-major configs[1] (bit-exact and FMA mode) in fresh processes, several times: the spread the staggered start leaves
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_cmrep; mkdir -p $O
for rep in 1 2 3 4; do
  for m in "" "--fused"; do
    ${PRE:-} timeout 300 python bench.py --workload biquad --layout chan $m --no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json - > $O/l.json 2> $O/l.err || tail -3 $O/l.err
    echo "rep $rep [$m]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-110)"
  done
done 2>&1 | tee $O/cm_repeat.log
