#!/bin/bash
# Round 5, GPU call 8: the driver's command with its full record kept (the final pass's stats step had overwritten it), then the
# rocprofv3 kernel trace of the same command on the same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05h
mkdir -p $O
cd $R
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" | head -1 | tee $O/smi.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json $O/bench_driver_full.json > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; python tools/show_line.py $O/bench_driver.json | cut -c1-200
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --full-json $O/bench_under_rocprof.json > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/stats -name "*kernel_trace.csv" -exec sh -c 'grep -E "Kernel_Name|alz::" "$1" > '$O'/kernel_dispatches.csv' _ {} \;
rm -rf $O/stats; head -6 $O/kernel_stats.csv | cut -c1-170
