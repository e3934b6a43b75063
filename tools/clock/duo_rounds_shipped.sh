R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_duo_rounds; mkdir -p $O
B="--no-cpu-baseline --no-secondary --steps 10 --warmup 3 --full-json -"
for cn in "16384 18" "24576 17" "32768 17" "8192 19" "65536 16" "12288 18"; do set -- $cn
  for m in "" "--fused"; do
  timeout 300 python bench.py $B --workload biquad $m --channels $1 --log2-samples $2 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "shipped biquad $1 ch x 2^$2 $m: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-170)"
  done
done 2>&1 | tee $O/shipped_rounds.log
timeout 900 python -m pytest tests/test_gpu_fullwidth.py -q -m gpu 2>&1 | tail -3
