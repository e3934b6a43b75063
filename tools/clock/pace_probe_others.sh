#!/bin/bash
# The common tile clock (alz_common.h pace_wait) tried on the other time-major streaming kernels, through the tuning build
# (make -C audiolazy_amd/csrc tuning): k_comb_tm (ALZ_COMB_PACE_GBPS), k_tvpc (ALZ_TVPC_PACE_GBPS), k_tvduo (ALZ_TVDUO_PACE_GBPS),
# k_mid (ALZ_MID_PACE_GBPS); 0 = free-running (the shipped form).  Then k_duo's FMA kernel at 2^16 samples (pace / no pace).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace_others; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
one() { # label, env, args
  env $(echo $2 | tr "," " ") timeout 300 python bench.py $B $3 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "$1 [$2]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-120)"
}
for rep in 1 2; do
  for g in 0 4400 4700 5000 5300 5600; do one "comb_fb time-major" ALZ_COMB_PACE_GBPS=$g "--workload comb"; done
  for g in 0 4600 4900 5200 5500 5800; do one "timevar per channel" ALZ_TVPC_PACE_GBPS=$g "--workload timevar --streams 0"; done
  for g in 0 3900 4200 4500 5000; do one "timevar shared" ALZ_TVDUO_PACE_GBPS=$g "--workload timevar"; done
  for g in 0 4800 5100 5400 5700; do one "maverage256" ALZ_MID_PACE_GBPS=$g "--workload maverage256"; done
  for g in 0 2300 2500 2800; do one "butter6" ALZ_MID_PACE_GBPS=$g "--workload butter6"; done
  for g in 0 5750; do one "biquad fma 2^16" ALZ_DUO_PACE_MIN_TILES=0,ALZ_DUO_PACE_GBPS=$g "--workload biquad --fused --log2-samples 16"; done
  for g in 0 5750; do one "biquad fma 2^15" ALZ_DUO_PACE_MIN_TILES=0,ALZ_DUO_PACE_GBPS=$g "--workload biquad --fused --log2-samples 15"; done
done 2>&1 | tee $O/pace_others.log
unset ALZ_LIBRARY
timeout 900 python -m pytest tests/test_gpu_fullwidth.py -q -m gpu -k "fused or cfg2" 2>&1 | tail -5 | tee $O/tests.log
