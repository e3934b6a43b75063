#!/bin/bash
# Round 5, call 36: the FMA mode keeps the default kernel where that one is faster (time-major, 256 - 320 two-wave workgroups):
# the width probe again, then the whole gpu suite on this library.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05aj
mkdir -p $O
cd $R
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
timeout 400 python tools/duo_fused_probe.py 2> $O/probe.err | tee $O/duo_fused.log | cut -c1-260
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
