#!/bin/bash
# a convoy (alz_common.h convoy_sync: sliding barrier over the workgroups, no rate to choose) in place of the tile clock, k_duo's clocked
# shapes; ALZ_CONVOY = Q | S << 8 (checkpoint every Q tiles, pass c when all have reached c - S); 0: the shipped clock
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_convoy; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
one() { # label, env, args
  env $(echo $2 | tr "," " ") timeout 120 python bench.py $B $3 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "$1 [$2]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
}
for rep in 1 2; do
  for cv in ${CONVOYS:-0 260 258 516 1028 264 520 1032 2064 272}; do   # 260 = Q 4 S 1; 258 = Q 2 S 1; 516 = Q 4 S 2; 1028 = Q 4 S 4; 264 = Q 8 S 1; 520 = Q 8 S 2; 1032 = Q 8 S 4; 2064 = Q 16 S 8; 272 = Q 16 S 1
    one "biquad fma" ALZ_CONVOY=$cv "--workload biquad --fused"
    one "envelope" ALZ_CONVOY=$cv "--workload envelope"
    one "biquad 8192 ch" ALZ_CONVOY=$cv "--workload biquad --channels 8192 --log2-samples 19"
  done
done 2>&1 | tee $O/convoy.log
