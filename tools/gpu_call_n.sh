#!/bin/bash
mkdir -p gpurun_out/r02n
export TMPDIR=/tmp
R=$(pwd)
b() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gs/s %s' % (d['value'], d['config']['parity_spot_check'][:40]))"; }
echo "shipped time: $(b)" | tee gpurun_out/r02n/nt.log
for v in tools/variants/wave_*.so; do
  echo "$(basename $v) time: $(ALZ_LIBRARY=$R/$v b)   chan: $(ALZ_LIBRARY=$R/$v b '--layout chan')" | tee -a gpurun_out/r02n/nt.log
done
echo "shipped time again: $(b)   chan: $(b '--layout chan')" | tee -a gpurun_out/r02n/nt.log
