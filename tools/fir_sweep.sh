#!/bin/bash
# A/B of the shared-tap FIR kernels on configs[2] (ALZ_FIR_OLD=1: k_fir<shared>, ALZ_FIR_S=1: k_fir_s, default: k_fir_ring)
run() { timeout 150 python bench.py --workload fir --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f %s %s' % (d['value'], d['config'].get('parity_spot_check'), d['config'].get('kernel')))"; }
echo "ring: $(run)"
echo "fir_s: $(ALZ_FIR_S=1 run)"
echo "old: $(ALZ_FIR_OLD=1 run)"
for f in tools/variants/fir_*.so; do [ -f $f ] && echo "$(basename $f): $(ALZ_LIBRARY=$PWD/$f run)"; done
timeout 300 python -m pytest tests -m gpu -x -q -k "fir" 2>&1 | tail -2
