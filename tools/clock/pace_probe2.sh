#!/bin/bash
# second pass of tools/clock/pace_probe_others.sh: where k_tvpc's and k_comb_tm's clocks stop being followed; block lengths
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace2; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
one() { # label, env, args
  env $(echo $2 | tr "," " ") timeout 300 python bench.py $B $3 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "$1 [$2]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-120)"
}
for rep in 1 2; do
  for g in 0 5800 6000 6200 6400 6600 6800 7200; do one "timevar per channel" ALZ_TVPC_PACE_GBPS=$g "--workload timevar --streams 0"; done
  for l in 14 16 17 19; do for g in 0 5800; do one "timevar per channel 2^$l" ALZ_TVPC_PACE_GBPS=$g "--workload timevar --streams 0 --log2-samples $l"; done; done
  for c in 2048 5120 8192; do for g in 0 5800; do one "timevar per channel $c ch" ALZ_TVPC_PACE_GBPS=$g "--workload timevar --streams 0 --channels $c --log2-samples 18"; done; done
  for g in 0 5600 5900 6200 6500; do one "comb_fb time-major" ALZ_COMB_PACE_GBPS=$g "--workload comb"; done
  for g in 0 5750; do one "biquad fma 2^17" ALZ_DUO_PACE_MIN_TILES=0,ALZ_DUO_PACE_GBPS=$g "--workload biquad --fused --log2-samples 17"; done
done 2>&1 | tee $O/pace2.log
unset ALZ_LIBRARY
timeout 900 python -m pytest tests/test_gpu_fullwidth.py -q -m gpu -k "fused or cfg2" 2>&1 | tail -5 | tee $O/tests.log
