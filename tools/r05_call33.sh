#!/bin/bash
# Round 5, call 33: the driver's command once more, now that profiles/r05_pmc_traffic_table.json carries the FIR rows of the
# shipped mapping (call 32 measured them; its own record still quoted the table of the interleaved mapping).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05ag
mkdir -p $O
cd $R
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json $O/bench_driver_full.json > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; python tools/show_line.py $O/bench_driver.json | cut -c1-170
