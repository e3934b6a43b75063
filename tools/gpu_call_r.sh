#!/bin/bash
# cfg4 k_pipe with direct global I/O for the first / last stage (time-major): parity, then A/B
mkdir -p gpurun_out/r02r
G="python bench.py --workload gammatone --no-cpu-baseline --steps 20 --warmup 3"
for v in base 1 2 3; do
  if [ $v = base ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$PWD/tools/variants/pipe_direct$v.so; fi
  timeout 600 python -m pytest tests/test_gpu_outer_narrow.py tests/test_gpu_bank.py tests/test_gpu_filters_api.py -x -q -m gpu 2>&1 | tail -2 > gpurun_out/r02r/pytest_$v.log
  cat gpurun_out/r02r/pytest_$v.log
  for lay in time chan; do
    timeout 200 $G --bank-layout $lay > gpurun_out/r02r/gt_${v}_$lay.json 2>/dev/null
    python - <<PY
import json
try:
  d=json.loads(open("gpurun_out/r02r/gt_${v}_$lay.json").read().strip().splitlines()[-1])
  print("$v $lay", round(d["value"],1), d["config"]["kernel"], d["config"]["parity_spot_check"][:40], d["roofline"]["frac"])
except Exception as e: print("$v $lay failed", e)
PY
  done
done
