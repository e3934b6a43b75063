#!/bin/bash
# rocprofv3 --kernel-trace --stats of the driver's command on the final code (kernel stats + dispatch rows)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --no-cpu-baseline --no-parity-check > $O/stats.log 2>&1
cd $R
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/stats -name "*kernel_trace.csv" -exec sh -c 'grep -E "Kernel_Name|alz::" "$1" > '$O'/kernel_dispatches.csv' _ {} \;
rm -rf $O/stats
head -14 $O/kernel_stats.csv | cut -c1-170
tail -1 $O/stats.log | cut -c1-300
