#!/bin/bash
# k_duo x-ring depth (tiles of DMA in flight): 4 (shipped) / 5 / 6 on the workloads whose REC wave is not the limit
mkdir -p gpurun_out/r02z
B="python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3"
for v in base duo_xr5 duo_xr6; do
  if [ $v = base ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$PWD/tools/variants/$v.so; fi
  for args in "" "--fused" "--workload envelope" "--layout chan --fused"; do
    timeout 200 $B $args 2>/dev/null > gpurun_out/r02z/b.json
    python - "$v" "$args" <<'PY'
import json,sys
try:
  d=json.loads(open("gpurun_out/r02z/b.json").read().strip().splitlines()[-1])
  print(sys.argv[1], sys.argv[2], round(d["value"],1), d["config"].get("kernel"), str(d["config"].get("parity_spot_check"))[:40], round(d["roofline"]["frac"],4))
except Exception as e: print(sys.argv[1], sys.argv[2], "failed", e)
PY
  done
done | tee gpurun_out/r02z/xring.log
