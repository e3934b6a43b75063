#!/bin/bash
# k_duo with AUX two tiles ahead (shipped) against the previous form (ALZ_DUO_AHEAD=0): parity first, then cfg2
mkdir -p gpurun_out/r02z
true
true
B="python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3"
for v in base duo_ahead0 base duo_ahead0; do
  if [ $v = base ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$PWD/tools/variants/$v.so; fi
  for args in "" "--layout chan" "--channels 2048" "--channels 512 --time-parallel 1"; do
    timeout 200 $B $args 2>/dev/null > gpurun_out/r02z/b.json
    python - "$v" "$args" <<'PY'
import json,sys
try:
  d=json.loads(open("gpurun_out/r02z/b.json").read().strip().splitlines()[-1])
  print(sys.argv[1], sys.argv[2], round(d["value"],1), d["config"].get("kernel"), str(d["config"].get("parity_spot_check"))[:40], round(d["roofline"]["frac"],4))
except Exception as e: print(sys.argv[1], sys.argv[2], "failed", e)
PY
  done
done | tee gpurun_out/r02z/ab.log
