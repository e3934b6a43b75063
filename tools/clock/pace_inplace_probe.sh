#!/bin/bash
# the tile clock on blocks processed IN PLACE (x == y: not "streaming" blocks, the kernels without non-temporal tiles): ALZ_DUO_PACE_ANY=1
# against 0 (the shipped rule: such blocks run free / keep round 4's paced pass); tools/variants/libalzhip_wave_tune.so
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace_inplace; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json - --in-place"
for rep in 1 2; do
  for a in "--workload envelope" "--workload biquad --channels 6144 --log2-samples 19" "--workload biquad --channels 8192 --log2-samples 19" "--workload biquad --fused" "--workload biquad --fused --channels 7168 --log2-samples 19" "--workload biquad"; do
    for any in 0 1; do
      ALZ_DUO_PACE_ANY=$any timeout 300 python bench.py $B $a > $O/l.json 2> $O/l.err || tail -3 $O/l.err
      echo "in place $a [clock on any big block: $any]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
    done
  done
done 2>&1 | tee $O/inplace.log
