#!/bin/bash
# fourth pass: the common tile clock in place of the one-pole banks' paced feed-forward pass (ALZ_DUO_AUXPACE=0); widths between one and
# two workgroups per CU, bit-exact and FMA; through tools/variants/libalzhip_wave_tune.so
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace4; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
one() { # label, env, args
  env $(echo $2 | tr "," " ") timeout 300 python bench.py $B $3 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "$1 [$2]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
}
for rep in 1 2; do
  one "envelope shipped" ALZ_WAVE_PACE_GBPS=0 "--workload envelope"
  for g in 5500 5700 5800 5900 6000 6100; do one "envelope" ALZ_DUO_AUXPACE=0,ALZ_WAVE_PACE_GBPS=$g "--workload envelope"; done
  for l in 16 17 18; do one "envelope 2^$l shipped" ALZ_WAVE_PACE_GBPS=0 "--workload envelope --log2-samples $l"; one "envelope 2^$l" ALZ_DUO_AUXPACE=0,ALZ_WAVE_PACE_GBPS=5700 "--workload envelope --log2-samples $l"; done
  one "envelope 5120 ch shipped" ALZ_WAVE_PACE_GBPS=0 "--workload envelope --channels 5120"
  for g in 5000 5300 5600; do one "envelope 5120 ch" ALZ_DUO_AUXPACE=0,ALZ_WAVE_PACE_GBPS=$g "--workload envelope --channels 5120"; done
  for c in 4608 5120 5632 6144 7168 7680; do
    for g in 0 4800 5200 5600; do one "biquad $c ch" ALZ_WAVE_PACE_GBPS=$g "--workload biquad --channels $c --log2-samples 19"; done
  done
  for c in 5120 7168; do
    for g in 0 5200 5600; do one "biquad $c ch fma" ALZ_DUO_PACE_GBPS=0,ALZ_WAVE_PACE_GBPS=$g "--workload biquad --fused --channels $c --log2-samples 19"; done
  done
done 2>&1 | tee $O/pace4.log
