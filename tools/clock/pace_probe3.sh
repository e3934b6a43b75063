#!/bin/bash
# third pass: the common tile clock on the bit-exact kernels of alz_wave.hip -- one-pole banks (envelope: |x| -> lowpass), wide banks
# (k_wave, 8192 - 16384 channels), through tools/variants/libalzhip_wave_tune.so (build_variant.sh wave_tune alz_wave.hip -DALZ_TUNING)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace3; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
one() { # label, env, args
  env $(echo $2 | tr "," " ") timeout 300 python bench.py $B $3 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "$1 [$2]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-120)"
}
for rep in 1 2; do
  for g in 0 5400 5600 5750 5900; do one "envelope (one pole)" ALZ_WAVE_PACE_GBPS=$g "--workload envelope"; done
  for g in 0 5600; do one "envelope, no paced pass" ALZ_DUO_AUXPACE=0,ALZ_WAVE_PACE_GBPS=$g "--workload envelope"; done
  for g in 0 5000 5300 5600 5900; do one "biquad 8192 ch" ALZ_WAVE_PACE_GBPS=$g "--workload biquad --channels 8192 --log2-samples 19"; done
  for g in 0 5000 5300 5600 5900; do one "biquad 16384 ch" ALZ_WAVE_PACE_GBPS=$g "--workload biquad --channels 16384 --log2-samples 18"; done
  for g in 0 5200 5600; do one "biquad 6144 ch" ALZ_WAVE_PACE_GBPS=$g "--workload biquad --channels 6144 --log2-samples 19"; done
  for g in 0 5200 5600; do one "biquad 6144 ch fma" ALZ_WAVE_PACE_GBPS=$g "--workload biquad --fused --channels 6144 --log2-samples 19"; done
  for g in 0 5300 5600; do one "biquad 8192 ch fma" ALZ_WAVE_PACE_GBPS=$g "--workload biquad --fused --channels 8192 --log2-samples 19"; done
done 2>&1 | tee $O/pace3.log
