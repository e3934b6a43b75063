#!/bin/bash
# Round 5, GPU call 3: the gpu suite after call 2's fixes, the time-parallel fuzzer (new) and the older GPU fuzzers, the
# lockstep-persistent copy experiment (is the one-shot grid's advantage the ORDER in which the chip walks the buffers?).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" | head -2 | tee $O/smi.log
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 > $O/pytest_gpu.log 2>&1; echo "pytest all rc=$?"; grep -n "FAILED\|passed\|failed" $O/pytest_gpu.log | tail -35 | cut -c1-250
timeout 600 python tools/fuzz_timeparallel.py 160 901 > $O/fuzz_timeparallel.log 2>&1; echo "fuzz_timeparallel rc=$?"; tail -25 $O/fuzz_timeparallel.log | cut -c1-600
for f in "fuzz_bank.py 150 504" "fuzz_outer.py 100 505"; do
  set -- $f
  timeout 300 python tools/$1 $2 $3 > $O/${1%.py}.log 2>&1; echo "$1 rc=$? $(tail -1 $O/${1%.py}.log | cut -c1-300)"
done
timeout 200 tools/ubench_copy2 4096 > $O/copy2_4g.log 2>&1; echo "copy2 rc=$?"; grep -v "^$" $O/copy2_4g.log | cut -c1-140
timeout 200 tools/ubench_copy2 1024 > $O/copy2_1g.log 2>&1; grep "k_lock\|one-shot grid, 1 x\|k_pers, 1 x\|k_pers, 8 x" $O/copy2_1g.log | cut -c1-140
