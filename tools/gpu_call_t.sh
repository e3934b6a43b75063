#!/bin/bash
# cfg4 k_pipe: hand-over writes spread through the interval (ALZ_PIPE_EARLYW), with / without direct input
mkdir -p gpurun_out/r02t
G="python bench.py --workload gammatone --no-cpu-baseline --steps 20 --warmup 3"
for v in ${VARIANTS:-base pipe_direct5 pipe_d0_e1 pipe_d5_e1}; do
  if [ $v = base ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$PWD/tools/variants/$v.so; fi
  for lay in time chan; do
    for fm in "" "--fused"; do
    timeout 200 $G --bank-layout $lay $fm > gpurun_out/r02t/gt_${v}_$lay$fm.json 2>/dev/null
    python - <<PY
import json
try:
  d=json.loads(open("gpurun_out/r02t/gt_${v}_$lay$fm.json").read().strip().splitlines()[-1])
  print("$v $lay $fm", round(d["value"],1), d["config"]["kernel"], d["config"]["parity_spot_check"][:40], round(d["roofline"]["frac"],4))
except Exception as e: print("$v $lay failed", e)
PY
    done
  done
done
