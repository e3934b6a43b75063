#!/bin/bash
# Round 5, call 37: where the dispatcher should hand a biquad bank from the two-wave kernel to the single-wave one
# (tools/duo_width_probe.py, -DALZ_TUNING build).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05ak
mkdir -p $O
cd $R
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
timeout 300 python tools/duo_width_probe.py 2>> $O/probe.err | tee $O/duo_width.log | cut -c1-200
ALZ_DUO_MAX_LANES=1000000 timeout 300 python tools/duo_width_probe.py 2>> $O/probe.err | tee -a $O/duo_width.log | cut -c1-200
