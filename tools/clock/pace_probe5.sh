#!/bin/bash
# fifth pass (the rule of launch_wave_impl in place): where the one-pole banks' clock stops being followed; 480 and 512 groups;
# k_duo instead of k_wave<16> at 8192 channels; then the shipped library on the shapes the rule covers, and the tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace5; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
one() { # label, env, args
  env $(echo $2 | tr "," " ") timeout 300 python bench.py $B $3 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "$1 [$2]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
}
for rep in 1 2; do
  for g in 0 5900 6100 6300 6500 6700 6900 7200; do one "envelope" ALZ_DUO_PACE_GBPS=$g "--workload envelope"; done
  for g in 5600 5900 6200; do one "biquad 7680 ch" ALZ_DUO_PACE_GBPS=$g "--workload biquad --channels 7680 --log2-samples 19"; done
  one "biquad 8192 ch k_wave<16>" X=0 "--workload biquad --channels 8192 --log2-samples 19"
  for g in 0 5600 5900 6200; do one "biquad 8192 ch k_duo" ALZ_DUO_MAX_LANES=16384,ALZ_DUO_PACE_GBPS=$g "--workload biquad --channels 8192 --log2-samples 19"; done
  for g in 0 5600 5900; do one "biquad 8192 ch k_duo fma" ALZ_DUO_MAX_LANES=16384,ALZ_DUO_PACE_GBPS=$g "--workload biquad --fused --channels 8192 --log2-samples 19"; done
done 2>&1 | tee $O/pace5.log
unset ALZ_LIBRARY
for a in "--workload envelope" "--workload biquad --channels 6144 --log2-samples 19" "--workload biquad --fused --channels 6144 --log2-samples 19" "--workload biquad --fused" "--workload biquad"; do
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 --full-json - $a > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "shipped $a: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-220)"
done 2>&1 | tee $O/shipped.log
timeout 1200 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_bank.py tests/test_gpu_filters_api.py -q -m gpu 2>&1 | tail -5 | tee $O/tests.log
