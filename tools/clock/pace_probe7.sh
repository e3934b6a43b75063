#!/bin/bash
# seventh pass: a clock that forgives (ALZ_PACE_FORGIVE = ticks: a wave further behind than that restarts its schedule from now instead of
# running free until it has caught up) against the fixed one (0), at the shipped rates and at rates the memory system does not follow
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace7; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
one() { # label, env, args
  env $(echo $2 | tr "," " ") timeout 300 python bench.py $B $3 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "$1 [$2]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
}
for rep in 1 2; do
  for fg in 0 8 40 150; do
    for g in 5500 5750 6000 6500 7500; do one "biquad fma" ALZ_PACE_FORGIVE=$fg,ALZ_DUO_PACE_GBPS=$g "--workload biquad --fused"; done
    for g in 5900 6300 7000; do one "envelope" ALZ_PACE_FORGIVE=$fg,ALZ_DUO_PACE_GBPS=$g "--workload envelope"; done
    for g in 5600 6200 7000; do one "timevar per channel" ALZ_PACE_FORGIVE=$fg,ALZ_TVPC_PACE_GBPS=$g "--workload timevar --streams 0"; done
  done
done 2>&1 | tee $O/pace7.log
