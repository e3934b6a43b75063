#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace_check; mkdir -p $O
B="--no-cpu-baseline --no-secondary --steps 10 --warmup 3 --full-json -"
for rep in 1 2; do
for a in "--workload biquad --fused" "--workload biquad --fused --no-parity-check" "--workload envelope --no-parity-check"; do
  timeout 300 python bench.py $B $a > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "shipped $a: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-120)"
  ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so timeout 300 python bench.py $B $a > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "tuning build, no env $a: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-120)"
done
done 2>&1 | tee $O/check.log
