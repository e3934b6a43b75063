#!/bin/bash
# in place, whole rounds of 512 workgroups: k_duo on the clock (ALZ_DUO_ROUNDS_ANY=1) against k_wave<16 / 64> (0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace_inplace; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json - --in-place"
for rep in 1 2; do
  for a in "--workload biquad --channels 8192 --log2-samples 19" "--workload biquad --channels 16384 --log2-samples 18" "--workload biquad --fused --channels 8192 --log2-samples 19" "--workload biquad --channels 32768 --log2-samples 17"; do
    for any in 0 1; do
      ALZ_DUO_ROUNDS_ANY=$any timeout 300 python bench.py $B $a > $O/l.json 2> $O/l.err || tail -3 $O/l.err
      echo "in place $a [k_duo in rounds on any big block: $any]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
    done
  done
done 2>&1 | tee $O/inplace2.log
