#!/bin/bash
# sixth pass: the clock with feedback (alz_common.h pace_wait, one slip word per launch) against the fixed clock, at the shipped rate and
# at rates the memory system does not follow; through the tuning build (ALZ_PACE_FEEDBACK=0: fixed clock)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace6; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
one() { # label, env, args
  env $(echo $2 | tr "," " ") timeout 300 python bench.py $B $3 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "$1 [$2]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
}
for rep in 1 2; do
  for fb in 1 0; do
    for g in 5750 5900 6100 6500 7500; do one "biquad fma" ALZ_PACE_FEEDBACK=$fb,ALZ_DUO_PACE_GBPS=$g "--workload biquad --fused"; done
    for g in 5900 6100 6300 7000; do one "envelope" ALZ_PACE_FEEDBACK=$fb,ALZ_DUO_PACE_GBPS=$g "--workload envelope"; done
    for g in 5600 5800 6200 7000; do one "timevar per channel" ALZ_PACE_FEEDBACK=$fb,ALZ_TVPC_PACE_GBPS=$g "--workload timevar --streams 0"; done
    for g in 5900 6500 7500; do one "comb_fb" ALZ_PACE_FEEDBACK=$fb,ALZ_COMB_PACE_GBPS=$g "--workload comb"; done
    for g in 5600 6200 7000; do one "biquad 7680 ch" ALZ_PACE_FEEDBACK=$fb,ALZ_DUO_PACE_GBPS=$g "--workload biquad --channels 7680 --log2-samples 19"; done
    for g in 5900 6500 7500; do one "biquad 8192 ch k_duo" ALZ_PACE_FEEDBACK=$fb,ALZ_DUO_MAX_LANES=16384,ALZ_DUO_PACE_GBPS=$g "--workload biquad --channels 8192 --log2-samples 19"; done
    for g in 5000 5600 6500; do one "biquad 5120 ch" ALZ_PACE_FEEDBACK=$fb,ALZ_DUO_PACE_GBPS=$g "--workload biquad --channels 5120 --log2-samples 19"; done
  done
done 2>&1 | tee $O/pace6.log
unset ALZ_LIBRARY
timeout 1200 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_bank.py tests/test_gpu_tv.py -q -m gpu 2>&1 | tail -5 | tee $O/tests.log
