#!/bin/bash
# the shipped library on blocks processed in place (non-temporal tiles where launch_wave_impl's nt_in_place says so), then the tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_nt_inplace; mkdir -p $O
B="--no-cpu-baseline --no-secondary --steps 10 --warmup 3 --full-json - --in-place --no-parity-check"
for rep in 1 2; do
for a in "--workload biquad" "--workload biquad --fused" "--workload envelope" "--workload biquad --channels 8192 --log2-samples 19" "--workload biquad --channels 16384 --log2-samples 18" "--workload biquad --channels 6144 --log2-samples 19" "--workload biquad --layout chan" "--workload biquad --layout chan --fused" "--workload biquad --channels 5120 --log2-samples 19"; do
  timeout 300 python bench.py $B $a > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "shipped, in place $a: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
done
done 2>&1 | tee $O/nt_inplace_shipped.log
timeout 1500 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_bank.py tests/test_gpu_filters_api.py tests/test_gpu_formats.py tests/test_gpu_host_path.py -q -m gpu 2>&1 | tail -3
