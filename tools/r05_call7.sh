#!/bin/bash
# Round 5, GPU call 7: the channel-major broadcast replay storing two tiles at a time (256-byte pieces per row) against
# tile-by-tile stores (tools/variants/libalzhip_nopair.so), after the parity tests of the cascade forms.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05g
mkdir -p $O
cd $R
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" | head -1 | tee $O/smi.log
timeout 400 python -m pytest -m gpu -q -x --timeout=300 tests/test_gpu_cscan_dot.py tests/test_gpu_outer_narrow.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-300
GT="--workload gammatone --streams 1 --log2-samples 20 --time-parallel 1 --no-cpu-baseline --steps 40 --warmup 5 --bank-layout chan"
for rep in 1 2 3; do
  for lib in "" nopair; do
    env ${lib:+ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so} timeout 300 python bench.py $GT > $O/gt_${lib:-pair}_$rep.json 2> $O/gt_${lib:-pair}_$rep.err
    echo "one stream TP chan [${lib:-pair}]: $(python tools/show_line.py $O/gt_${lib:-pair}_$rep.json | head -1 | cut -c1-150)"
  done
done
timeout 200 python tools/fuzz_timeparallel.py 40 77 > $O/fuzz.log 2>&1; echo "fuzz rc=$? $(tail -1 $O/fuzz.log | cut -c1-250)"
