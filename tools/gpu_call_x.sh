#!/bin/bash
# shipped build after the k_pipe changes: cascade / bank / gammatone parity, then the one-stream and cfg4 lines
mkdir -p gpurun_out/r02x
timeout 900 python -m pytest tests/test_gpu_outer_narrow.py tests/test_gpu_fullwidth.py tests/test_gpu_bank.py tests/test_gpu_filters_api.py tests/test_gpu_scan.py -x -q -m gpu > gpurun_out/r02x/pytest.log 2>&1
tail -4 gpurun_out/r02x/pytest.log
for lay in chan time; do
  timeout 200 python bench.py --workload gammatone --no-cpu-baseline --steps 20 --warmup 3 --bank-layout $lay 2>/dev/null | tee gpurun_out/r02x/gt_$lay.json | cut -c1-400
done
