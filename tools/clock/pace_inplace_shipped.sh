R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace_inplace; mkdir -p $O
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json - --in-place"
for a in "--workload envelope" "--workload biquad --channels 6144 --log2-samples 19" "--workload biquad --channels 8192 --log2-samples 19" "--workload biquad --channels 16384 --log2-samples 18" "--workload biquad"; do
  timeout 300 python bench.py $B $a > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "shipped, in place $a: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
done 2>&1 | tee $O/inplace_shipped.log
timeout 1200 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_bank.py tests/test_gpu_filters_api.py tests/test_gpu_formats.py -q -m gpu 2>&1 | tail -3
