#!/bin/bash
# Round 5, GPU call 2: k_look for channel-major / |x| / in-place blocks (new tests first, then the suite), the driver's line,
# a kernel trace of the one-stream gammatone time-parallel workload in both layouts, and two A/B libraries
# (tools/variants: k_wave with non-temporal tiles on wide banks; k_cdot2 = hand-placed operand requests).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
export TMPDIR=/tmp
rocm-smi --showuniqueid --showserial --showpower --showclocks 2>/dev/null | grep -v "^=\|^$" | head -20 > $O/smi.log; head -8 $O/smi.log
timeout 600 python -m pytest -m gpu -q -x --timeout=300 tests/test_gpu_scan.py tests/test_gpu_outer_narrow.py tests/test_gpu_maps.py tests/test_gpu_cscan_dot.py > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -25 $O/pytest_new.log | cut -c1-300
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json $O/bench_full.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/show_line.py $O/bench.json 2>&1 | tail -30
GT="--workload gammatone --streams 1 --log2-samples 20 --time-parallel 1 --no-cpu-baseline --steps 20 --warmup 3"
for lay in chan time; do
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$lay -o t -- python $R/bench.py $GT --bank-layout $lay --no-parity-check > $O/trace_$lay.log 2>&1 )
  find $O/trace_$lay -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_gt_$lay.csv \;
  echo "-- kernel stats, one-stream gammatone time-parallel, layout $lay"; head -8 $O/kernel_stats_gt_$lay.csv | cut -c1-200
  rm -rf $O/trace_$lay
done
for lib in "" cdot2; do
  for lay in chan time; do
    env ${lib:+ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so} timeout 300 python bench.py $GT --bank-layout $lay > $O/gt_${lib:-ship}_$lay.json 2> $O/gt_${lib:-ship}_$lay.err
    echo "gammatone one stream TP [${lib:-ship}] [$lay]: $(python tools/show_line.py $O/gt_${lib:-ship}_$lay.json | head -1 | cut -c1-150)"
  done
done
for lib in "" wavent; do
  for args in "--channels 65536 --log2-samples 16 --layout time" "--channels 65536 --log2-samples 16 --layout chan" "--channels 16384 --log2-samples 18 --layout time"; do
    env ${lib:+ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so} timeout 300 python bench.py $args --no-secondary --no-cpu-baseline --steps 20 --warmup 3 > $O/wide.json 2> $O/wide.err
    echo "wide bank [${lib:-ship}] [$args]: $(python tools/show_line.py $O/wide.json | head -1 | cut -c1-150)"
  done
done
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 > $O/pytest_gpu.log 2>&1; echo "pytest all rc=$?"; tail -8 $O/pytest_gpu.log | cut -c1-300
