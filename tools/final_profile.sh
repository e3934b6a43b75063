#!/bin/bash
# Round-end measurement pass on the GPU box: bench lines for every workload, the rocprofv3 kernel
# stats of the contract command, and the two PMC passes behind roofline.traffic (--light: bench
# lines and kernel stats only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 300 python bench.py > $O/bench_full.json.log 2>$O/bench_full.err
timeout 200 python bench.py --layout chan --no-cpu-baseline > $O/bench_chan.json.log 2>/dev/null
timeout 200 python bench.py --workload fir --no-cpu-baseline > $O/bench_fir.json.log 2>/dev/null
timeout 200 python bench.py --workload gammatone --no-cpu-baseline > $O/bench_gammatone.json.log 2>/dev/null
timeout 200 python bench.py --workload lpc --no-cpu-baseline > $O/bench_lpc.json.log 2>/dev/null
if [ "$1" != "--light" ]; then
timeout 200 python tools/io_time.py > $O/io_time.log 2>&1
timeout 200 python tools/tv_time.py > $O/tv_time.log 2>&1
fi
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --no-cpu-baseline --no-parity-check > $O/stats.log 2>&1
cd $R
if [ "$1" != "--light" ]; then
timeout 400 bash tools/pmc_run.sh final_fetch FETCH_SIZE -- python $R/bench.py --log2-samples 18 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check
timeout 400 bash tools/pmc_run.sh final_write WRITE_SIZE -- python $R/bench.py --log2-samples 18 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check
python tools/pmc_traffic.py gpurun_out/pmc_final_fetch gpurun_out/pmc_final_write k_duo 4096 262144 > $O/pmc_traffic.json 2>$O/pmc_traffic.err
fi
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/stats -name "*kernel_trace.csv" -exec sh -c 'grep -E "Kernel_Name|k_duo" "$1" | cut -d, -f1-20 > '$O'/kernel_dispatches.csv' _ {} \;
rm -rf $O/stats
ls -la $O
