#!/bin/bash
# Round 5, GPU call 4: the chunk-major broadcast replay for one-stream channel-major blocks (k_casc<bc>), the probe fix
# (fuzzer again), SQ / TCP counters of k_cdot.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" | head -1 | tee $O/smi.log
timeout 600 python -m pytest -m gpu -q -x --timeout=300 tests/test_gpu_cscan_dot.py tests/test_gpu_outer_narrow.py tests/test_gpu_covariance.py tests/test_gpu_scan.py > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -12 $O/pytest_new.log | cut -c1-300
timeout 500 python tools/fuzz_timeparallel.py 120 902 > $O/fuzz_timeparallel.log 2>&1; echo "fuzz_timeparallel rc=$?"; tail -6 $O/fuzz_timeparallel.log | cut -c1-700
GT="--workload gammatone --streams 1 --log2-samples 20 --time-parallel 1 --no-cpu-baseline --steps 40 --warmup 5"
for lay in chan time; do
  timeout 300 python bench.py $GT --bank-layout $lay > $O/gt_$lay.json 2> $O/gt_$lay.err
  echo "gammatone one stream TP [$lay]: $(python tools/show_line.py $O/gt_$lay.json | head -1 | cut -c1-200)"
done
bash tools/gpu_call.sh r05d "pmc:SQ_WAVE_CYCLES+SQ_BUSY_CYCLES+SQ_ACTIVE_INST_VALU+SQ_ACTIVE_INST_SCA+SQ_WAIT_INST_ANY+SQ_WAIT_ANY+SQ_ACTIVE_INST_ANY:--workload+gammatone+--streams+1+--log2-samples+20+--time-parallel+1+--no-cpu-baseline+--no-parity-check+--steps+6+--warmup+2" \
  "pmc:SQ_INSTS_VALU+SQ_INSTS_SMEM+SQ_INSTS_VMEM_RD+SQ_INSTS_LDS+SQ_WAIT_INST_LDS+TCP_TOTAL_CACHE_ACCESSES_sum+TCP_TCC_READ_REQ_sum:--workload+gammatone+--streams+1+--log2-samples+20+--time-parallel+1+--no-cpu-baseline+--no-parity-check+--steps+6+--warmup+2" 2>&1 | grep -A12 "k_cdot\|k_casc\|k_cscan_fix" | cut -c1-160 | head -80
