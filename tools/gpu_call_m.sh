#!/bin/bash
mkdir -p gpurun_out/r02m
export TMPDIR=/tmp
timeout 600 python tools/fuzz_bank.py > gpurun_out/r02m/fuzz_bank.log 2>&1; tail -3 gpurun_out/r02m/fuzz_bank.log
timeout 600 python tools/fuzz_stream.py > gpurun_out/r02m/fuzz_stream.log 2>&1; tail -3 gpurun_out/r02m/fuzz_stream.log
