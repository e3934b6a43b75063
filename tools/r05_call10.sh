#!/bin/bash
# Round 5, GPU call 10: the tree as the driver will find it -- the gpu suite, smoke(), and LDS bank-conflict counters of
# k_look in both layouts (the channel-major form is new: its LOAD wave reads k_duo's channel-major DMA image).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for lay in time chan; do
  bash tools/gpu_call.sh r05j "pmc:SQ_LDS_BANK_CONFLICT+SQ_LDS_IDX_ACTIVE+SQ_INSTS_LDS+SQ_WAIT_INST_LDS+SQ_ACTIVE_INST_LDS+SQ_WAVE_CYCLES+SQ_BUSY_CYCLES:--channels+512+--time-parallel+1+--layout+$lay+--no-secondary+--no-cpu-baseline+--no-parity-check+--steps+6+--warmup+2" 2>&1 | grep -A9 "k_look" | cut -c1-150 | head -12
done
