#!/bin/bash
# k_pipe with two workgroups per CU (ALZ_PIPE_TWO) against the shipped build, 64 / 128 / 256 streams
mkdir -p gpurun_out/r02u
G="python bench.py --workload gammatone --no-cpu-baseline --steps 10 --warmup 2"
for v in base pipe_two; do
  if [ $v = base ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$PWD/tools/variants/$v.so; fi
  for st in 64 128 256; do
    for fm in "" "--fused"; do
    timeout 200 $G --streams $st $fm > gpurun_out/r02u/gt_${v}_$st$fm.json 2>/dev/null
    python - <<PY
import json
try:
  d=json.loads(open("gpurun_out/r02u/gt_${v}_$st$fm.json").read().strip().splitlines()[-1])
  print("$v $st $fm", round(d["value"],1), d["config"]["kernel"], d["config"]["parity_spot_check"][:40], round(d["roofline"]["frac"],4))
except Exception as e: print("$v $st failed", e)
PY
    done
  done
done
