#!/bin/bash
# few-input OUTER banks: parity, then the one-stream gammatone bank in its three modes
mkdir -p gpurun_out/r02q
timeout 900 python -m pytest tests/test_gpu_outer_narrow.py tests/test_gpu_bank.py tests/test_gpu_fullwidth.py -x -q -m gpu > gpurun_out/r02q/pytest.log 2>&1
tail -5 gpurun_out/r02q/pytest.log
for tp in 0 1; do
  timeout 300 python bench.py --workload gammatone --streams 1 --log2-samples 20 --time-parallel $tp --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r02q/gt1_tp$tp.json 2>gpurun_out/r02q/gt1_tp$tp.err
  cat gpurun_out/r02q/gt1_tp$tp.json
done
timeout 300 python bench.py --workload gammatone --streams 1 --log2-samples 20 --fused --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r02q/gt1_fma.json 2>&1
cat gpurun_out/r02q/gt1_fma.json
timeout 300 python bench.py --workload gammatone --streams 4 --log2-samples 18 --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r02q/gt4.json 2>&1
cat gpurun_out/r02q/gt4.json
