#!/bin/bash
mkdir -p gpurun_out/r02fuzz
timeout 600 python tools/fuzz_outer.py 250 > gpurun_out/r02fuzz/fuzz_outer.log 2>&1; echo "fuzz_outer rc=$?"; tail -4 gpurun_out/r02fuzz/fuzz_outer.log
timeout 600 python tools/fuzz_bank.py 300 77 > gpurun_out/r02fuzz/fuzz_bank.log 2>&1; echo "fuzz_bank rc=$?"; tail -2 gpurun_out/r02fuzz/fuzz_bank.log
timeout 600 python tools/fuzz_stream.py > gpurun_out/r02fuzz/fuzz_stream.log 2>&1; echo "fuzz_stream rc=$?"; tail -2 gpurun_out/r02fuzz/fuzz_stream.log
timeout 600 python -m pytest tests/test_gpu_outer_narrow.py tests/test_gpu_fullwidth.py tests/test_gpu_filters_api.py -x -q -m gpu 2>&1 | tail -2
