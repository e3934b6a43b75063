#!/bin/bash
# Round 5, call 38: smoke() and the driver's command on HEAD (after the FMA mode's dispatch rule; the suite ran in call 36).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05al
mkdir -p $O
cd $R
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json $O/bench_driver_full.json > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; python tools/show_line.py $O/bench_driver.json | cut -c1-170
