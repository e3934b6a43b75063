#!/bin/bash
# cfg4 k_pipe direct input, channel-major variant (5 = time-major + channel-major direct input); FMA mode too
mkdir -p gpurun_out/r02s
G="python bench.py --workload gammatone --no-cpu-baseline --steps 20 --warmup 3"
for v in base 5; do
  if [ $v = base ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$PWD/tools/variants/pipe_direct$v.so; fi
  if [ $v != base ]; then
    timeout 600 python -m pytest tests/test_gpu_outer_narrow.py tests/test_gpu_fullwidth.py tests/test_gpu_filters_api.py -x -q -m gpu 2>&1 | tail -2 > gpurun_out/r02s/pytest_$v.log
    cat gpurun_out/r02s/pytest_$v.log
  fi
  for lay in time chan; do
    for fm in "" "--fused"; do
    timeout 200 $G --bank-layout $lay $fm > gpurun_out/r02s/gt_${v}_$lay$fm.json 2>/dev/null
    python - <<PY
import json
try:
  d=json.loads(open("gpurun_out/r02s/gt_${v}_$lay$fm.json").read().strip().splitlines()[-1])
  print("$v $lay $fm", round(d["value"],1), d["config"]["kernel"], d["config"]["parity_spot_check"][:40], round(d["roofline"]["frac"],4))
except Exception as e: print("$v $lay failed", e)
PY
    done
  done
done
