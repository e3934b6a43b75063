#!/bin/bash
# k_pipe A/B on configs[3]: ALZ_PIPE 1/2 (sections per stage wave), shipped build vs the variants under
# tools/variants/ (built with -DALZ_PIPE_DIRECT=0 / -DALZ_PIPE_OVERLAP=0), ablations (ALZ_WAVE_DEBUG bits:
# 1 no DMA, 2 no arithmetic, 4 no stores)
run() { timeout 100 python bench.py --workload gammatone --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f %s' % (d['value'], d['config'].get('parity_spot_check')))"; }
for pipe in 1 2; do
echo "shipped pipe=$pipe: $(ALZ_PIPE=$pipe run)"
for f in tools/variants/casc_*.so; do [ -f $f ] && echo "$(basename $f) pipe=$pipe: $(ALZ_LIBRARY=$PWD/$f ALZ_PIPE=$pipe run)"; done
done
for d in 1 2 4 7; do echo "shipped pipe=1 dbg=$d: $(ALZ_PIPE=1 ALZ_WAVE_DEBUG=$d run)"; done
timeout 300 python -m pytest tests -m gpu -x -q -k "pipe or casc or gammatone or cascade" 2>&1 | tail -3
