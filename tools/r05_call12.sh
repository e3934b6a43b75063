#!/bin/bash
# Round 5, GPU call 12: k_casc<bc> with its sample requests hidden from hipcc's wait-count pass (it had been draining the
# wave's own stores every tile): parity of the cascade forms, the fuzzer, the one-stream bank in both layouts, a trace.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05l
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 400 python -m pytest -m gpu -q -x --timeout=300 tests/test_gpu_cscan_dot.py tests/test_gpu_outer_narrow.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -2
timeout 300 python tools/fuzz_timeparallel.py 60 1303 > $O/fuzz.log 2>&1; echo "fuzz rc=$? $(tail -1 $O/fuzz.log | cut -c1-200)"
GT="--workload gammatone --streams 1 --log2-samples 20 --time-parallel 1 --no-cpu-baseline --steps 40 --warmup 5"
for rep in 1 2; do
  for lay in chan time; do
    timeout 300 python bench.py $GT --bank-layout $lay > $O/gt_${lay}_$rep.json 2> /dev/null
    echo "one stream TP [$lay]: $(python tools/show_line.py $O/gt_${lay}_$rep.json | head -1 | cut -c1-150)"
  done
done
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py $GT --bank-layout chan --no-parity-check > $O/trace.log 2>&1 )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/trace; head -5 $O/kernel_stats.csv | cut -c1-180
