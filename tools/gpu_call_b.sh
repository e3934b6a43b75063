#!/bin/bash
mkdir -p gpurun_out/r02b
export TMPDIR=/tmp
python tools/scan_debug.py > gpurun_out/r02b/scan_debug.log 2>&1
ALZ_G=16 python tools/scan_debug.py > gpurun_out/r02b/scan_debug_g16.log 2>&1
cat gpurun_out/r02b/scan_debug.log gpurun_out/r02b/scan_debug_g16.log
# mid-width dispatcher sweep
for C in 4096 6144 8192 12288 16384; do
  for cfg in "ALZ_G=16" "ALZ_G=16 ALZ_DUO=0" "ALZ_G=32" "ALZ_G=64" ""; do
    L=$((34 - $(python3 -c "import math;print(int(math.ceil(math.log2($C))))")))
    r=$(env $cfg python bench.py --channels $C --log2-samples $L --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-parity-check 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %s' % (d['value'], d['config']['kernel']))")
    echo "channels=$C log2n=$L [$cfg] $r" | tee -a gpurun_out/r02b/width_sweep.log
  done
done
