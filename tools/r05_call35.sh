#!/bin/bash
# Round 5, call 35: where the opt-in FMA mode of a biquad bank helps and where it does not (tools/duo_fused_probe.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05ai
mkdir -p $O
cd $R
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
timeout 400 python tools/duo_fused_probe.py 2> $O/probe.err | tee $O/duo_fused.log | cut -c1-300
tail -3 $O/probe.err
