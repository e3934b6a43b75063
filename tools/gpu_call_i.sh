#!/bin/bash
mkdir -p gpurun_out/r02i
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_filters_api.py tests/test_gpu_timevar.py tests/test_gpu_reference_tests.py tests/test_gpu_reference_tests2.py tests/test_gpu_bank.py tests/test_gpu_maps.py -q -x > gpurun_out/r02i/pytest.log 2>&1
tail -4 gpurun_out/r02i/pytest.log
python tools/protocol_time.py 2>/dev/null | tee gpurun_out/r02i/protocol.log
