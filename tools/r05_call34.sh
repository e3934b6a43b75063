#!/bin/bash
# Round 5, call 34: two FIR launches sharing the chip (two streams) -- what the chains' pacing costs when a chain's waves are
# not resident together (tools/fir_two_streams.py); shipped mapping against the interleaved one, -DALZ_TUNING build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05ah
mkdir -p $O
cd $R
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
timeout 200 python tools/fir_two_streams.py 2>&1 | tee $O/two_streams.log | cut -c1-400
timeout 200 python tools/fir_two_streams.py fma 2>&1 | tee -a $O/two_streams.log | cut -c1-400
