#!/bin/bash
# Round 5, GPU call 1: the driver's command as the FIRST GPU process of the lease (lpc_bit_identical's 57 us), the fresh-box
# LPC timing probe, the gpu suite (new: tests/test_gpu_cscan_dot.py), the HBM ceiling sweep.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json $O/bench_full.json > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
echo "bench rc=$?"; python tools/show_line.py $O/bench.json 2>&1 | tail -40
timeout 300 python tools/lpc_fresh.py > $O/lpc_fresh.log 2>&1; echo "lpc_fresh rc=$?"; cat $O/lpc_fresh.log | cut -c1-400
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -x --timeout=300 tests/test_gpu_cscan_dot.py > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -40 $O/pytest_new.log
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 > $O/pytest_gpu.log 2>&1; echo "pytest all rc=$?"; tail -15 $O/pytest_gpu.log
timeout 300 tools/ubench_copy2 > $O/copy2.log 2>&1; echo "copy2 rc=$?"; cat $O/copy2.log
