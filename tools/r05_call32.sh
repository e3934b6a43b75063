#!/bin/bash
# Round 5, call 32 (last): HEAD with k_fir_ring's chains as the driver will find it -- the gpu suite, smoke(), the driver's
# command with its full record, the kernel trace of the same command, FETCH_SIZE / WRITE_SIZE of the two FIR workloads.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05af
mkdir -p $O
cd $R
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json $O/bench_driver_full.json > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; python tools/show_line.py $O/bench_driver.json | cut -c1-170
bash tools/pmc_workloads.sh r05af/pmc fir256 > $O/pmc_workloads.log 2>&1; tail -3 $O/pmc_workloads.log | cut -c1-200
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --full-json $O/bench_under_rocprof.json > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/stats -name "*kernel_trace.csv" -exec sh -c 'grep -E "Kernel_Name|alz::" "$1" > '$O'/kernel_dispatches.csv' _ {} \;
rm -rf $O/stats; grep "k_fir_ring" $O/kernel_stats.csv | cut -c1-170
