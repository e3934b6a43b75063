#!/bin/bash
mkdir -p gpurun_out/r02o
export TMPDIR=/tmp
g() { python bench.py --workload gammatone --steps 10 --warmup 2 $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gs/s %s %s' % (d['value'], d['config']['kernel'], d['config']['parity_spot_check'][:80]))"; }
echo "fma, one section per wave: $(g --fused)" | tee gpurun_out/r02o/gammatone.log
echo "fma, two sections per wave (ALZ_PIPE=2): $(ALZ_PIPE=2 g --fused)" | tee -a gpurun_out/r02o/gammatone.log
echo "exact, two sections per wave (ALZ_PIPE=2): $(ALZ_PIPE=2 g)" | tee -a gpurun_out/r02o/gammatone.log
