#!/bin/bash
# Round 5, call 42: launch_fir's gate again requires the chip to hold a whole number of waves of every channel group (call
# 41: 10 240 channels took chains of 4 and lost 6 %); the probe on that width, then the whole gpu suite.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05ap
mkdir -p $O
cd $R
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so timeout 200 python tools/fir_map_probe.py --configs map1,auto --channels 10240 --rows 209664 2>> $O/probe.err | tee $O/probe_10240.log | cut -c1-250
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
