#!/bin/bash
# the two-pole bit-exact bank of 4096 channels IN PLACE (14.4 ms against 13.1 - 13.7 out of place: not at the recurrence wave's floor
# there): does a clock bring it back?  tools/variants/libalzhip_wave_tune.so, ALZ_DUO_PACE_GBPS overrides the rule's rate (0 for this shape)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace_inplace; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
for rep in 1 2 3; do
  timeout 300 python bench.py $B --workload biquad > $O/l.json 2> $O/l.err; echo "out of place [free]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-90)"
  for g in 0 4700 4900 5100 5300; do
    ALZ_DUO_PACE_GBPS=$g timeout 300 python bench.py $B --workload biquad --in-place > $O/l.json 2> $O/l.err || tail -3 $O/l.err
    echo "in place [clock $g]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-90)"
  done
done 2>&1 | tee $O/inplace3.log
