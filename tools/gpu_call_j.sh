#!/bin/bash
mkdir -p gpurun_out/r02j
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_timevar.py tests/test_gpu_reference_tests.py -q -x > gpurun_out/r02j/pytest.log 2>&1
tail -15 gpurun_out/r02j/pytest.log
echo "--- k_tvduo on" | tee gpurun_out/r02j/tv_time.log
timeout 120 python tools/tv_time.py 2>/dev/null | tee -a gpurun_out/r02j/tv_time.log
echo "--- ALZ_TV_DUO=0" | tee -a gpurun_out/r02j/tv_time.log
ALZ_TV_DUO=0 timeout 120 python tools/tv_time.py 2>/dev/null | tee -a gpurun_out/r02j/tv_time.log
