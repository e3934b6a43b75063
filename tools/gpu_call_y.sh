#!/bin/bash
# single-wave cascade (k_casc, ALZ_PIPE=0) against the wave pipeline on banks wide enough for one wave per SIMD
mkdir -p gpurun_out/r02y
for st in 64 256 512; do
  for pipe in 1 0; do
    for lay in chan time; do
    ALZ_PIPE=$pipe timeout 300 python bench.py --workload gammatone --no-cpu-baseline --steps 5 --warmup 2 --streams $st --bank-layout $lay > gpurun_out/r02y/gt_${st}_p${pipe}_$lay.json 2>/dev/null
    python - <<PY
import json
try:
  d=json.loads(open("gpurun_out/r02y/gt_${st}_p${pipe}_$lay.json").read().strip().splitlines()[-1])
  print("streams=$st ALZ_PIPE=$pipe $lay", round(d["value"],1), d["config"]["kernel"], d["config"]["parity_spot_check"][:30], round(d["roofline"]["frac"],4))
except Exception as e: print("streams=$st ALZ_PIPE=$pipe $lay failed", e)
PY
    done
  done
done | tee gpurun_out/r02y/wide.log
