#!/bin/bash
# non-temporal tiles (and with them the streaming-block rules: the FMA kernel with the storing wave, rounds of k_duo) for blocks processed
# IN PLACE: ALZ_NT_INPLACE=1 in a tuning build of alz_api.hip (tools/build_variant.sh api_tune alz_api.hip -DALZ_TUNING) against 0 (shipped)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_nt_inplace; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_api_tune.so
B="--no-cpu-baseline --no-secondary --steps 10 --warmup 3 --full-json - --in-place --no-parity-check"
for rep in 1 2; do
  for a in "--workload biquad" "--workload biquad --fused" "--workload envelope" "--workload biquad --channels 8192 --log2-samples 19" "--workload biquad --channels 16384 --log2-samples 18" "--workload biquad --channels 6144 --log2-samples 19" "--workload biquad --layout chan" "--workload biquad --layout chan --fused"; do
    for nt in 0 1; do
      ALZ_NT_INPLACE=$nt timeout 300 python bench.py $B $a > $O/l.json 2> $O/l.err || tail -3 $O/l.err
      echo "in place $a [non-temporal tiles in place: $nt]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
    done
  done
done 2>&1 | tee $O/nt_inplace.log
ALZ_NT_INPLACE=1 timeout 900 python -m pytest tests/test_gpu_fullwidth.py -q -m gpu -k "in_place" 2>&1 | tail -3
