#!/bin/bash
# k_duo, time-major: the helper wave's lane groups own 16 consecutive rows (18 LDS reads per tile) against rows 4 j + q (48 reads);
# default kernel and the FMA instantiations with the storing wave (variant builds of alz_wave.hip)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_duorows; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_bank.py tests/test_gpu_scan.py -x -q 2>&1 | tail -3
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
for rep in 1 2; do
  for cfg in "shipped:" "duo_rows0:" "duo_fma3r:--fused" "duo_fma3r0:--fused" "shipped:--fused --channels 2048" "duo_rows0:--fused --channels 2048" "shipped:--channels 2048" "duo_rows0:--channels 2048" "shipped:--channels 8192 --log2-samples 19" "duo_rows0:--channels 8192 --log2-samples 19"; do
    lib=${cfg%%:*}; m=${cfg#*:}
    if [ $lib = shipped ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so; fi
    timeout 300 python bench.py --workload biquad $m $B > $O/l.json 2> $O/l.err || tail -3 $O/l.err
    echo "$lib [$m]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-110)"
  done
done 2>&1 | tee $O/duo_rows.log
