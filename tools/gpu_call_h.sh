#!/bin/bash
mkdir -p gpurun_out/r02h
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_host_path.py tests/test_gpu_filters_api.py -q > gpurun_out/r02h/pytest.log 2>&1
tail -5 gpurun_out/r02h/pytest.log
echo "--- pipelined host path" | tee gpurun_out/r02h/host_path.log
python tools/host_path_time.py 2>/dev/null | tee -a gpurun_out/r02h/host_path.log
echo "--- ALZ_HOST_PIPE=0 (pageable, synchronous)" | tee -a gpurun_out/r02h/host_path.log
ALZ_HOST_PIPE=0 python tools/host_path_time.py 2>/dev/null | tee -a gpurun_out/r02h/host_path.log
./tools/ubench_mfma64 | tee gpurun_out/r02h/ubench_mfma64.log
