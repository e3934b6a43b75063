#!/bin/bash
# k_pipe timing ablations on configs[3] (ALZ_ABLATE build: results are wrong by design, only the time counts)
# bits: 2 = no section arithmetic, 4 = no stores, 8 = no barriers, 16 = no LDS drain before the barrier
mkdir -p gpurun_out/r02w
export ALZ_LIBRARY=$PWD/tools/variants/pipe_ablate.so
for dbg in 0 2 8 16 24 10 26 4 6 30; do
  for fm in "" "--fused"; do
  ALZ_WAVE_DEBUG=$dbg timeout 200 python bench.py --workload gammatone --no-cpu-baseline --no-parity-check --steps 10 --warmup 2 $fm > gpurun_out/r02w/gt_$dbg$fm.json 2>/dev/null
  python - <<PY
import json
try:
  d=json.loads(open("gpurun_out/r02w/gt_$dbg$fm.json").read().strip().splitlines()[-1])
  print("dbg=$dbg $fm", round(d["value"],1), "Gsamples/s", round(d["roofline"]["kernel_ms_avg"],3), "ms")
except Exception as e: print("dbg=$dbg failed", e)
PY
  done
done | tee gpurun_out/r02w/ablations.log
