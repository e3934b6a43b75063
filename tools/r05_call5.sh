#!/bin/bash
# Round 5, GPU call 5: k_cdot3 (whole-line row loads, half-step response pipelining) against k_cdot, same library (the tuning
# build reads ALZ_CDOT_V3): parity first (the cascade tests through the variant), then both layouts of the one-stream bank,
# a kernel trace, and the chunk-length sweep.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" | head -1 | tee $O/smi.log
V=$R/tools/variants/libalzhip_tuning.so
ALZ_CDOT_V3=1 ALZ_LIBRARY=$V timeout 400 python -m pytest -m gpu -q -x --timeout=300 tests/test_gpu_cscan_dot.py tests/test_gpu_outer_narrow.py > $O/pytest_v3.log 2>&1; echo "pytest (k_cdot3) rc=$?"; tail -6 $O/pytest_v3.log | cut -c1-300
GT="--workload gammatone --streams 1 --log2-samples 20 --time-parallel 1 --no-cpu-baseline --steps 40 --warmup 5"
for v3 in 0 1 0 1; do
  for lay in chan time; do
    ALZ_CDOT_V3=$v3 ALZ_LIBRARY=$V timeout 300 python bench.py $GT --bank-layout $lay > $O/gt_${v3}_$lay.json 2> $O/gt_${v3}_$lay.err
    echo "one stream TP [k_cdot3=$v3] [$lay]: $(python tools/show_line.py $O/gt_${v3}_$lay.json | head -1 | cut -c1-150)"
  done
done
( cd /tmp; ALZ_CDOT_V3=1 ALZ_LIBRARY=$V timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py $GT --bank-layout chan --no-parity-check > $O/trace.log 2>&1 )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_v3.csv \; ; rm -rf $O/trace; head -6 $O/kernel_stats_v3.csv | cut -c1-180
ALZ_CDOT_V3=1 ALZ_LIBRARY=$V timeout 300 python tools/tp_chunk_sweep.py > $O/chunk_sweep.log 2>&1; cat $O/chunk_sweep.log | cut -c1-160
