#!/bin/bash
# last validation of the round: whole GPU suite, smoke(), the driver's bench command (set SKIP_TESTS=1 for the bench alone)
mkdir -p gpurun_out/final2
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final2/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/final2/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
fi
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/final2/bench_full.json 2> gpurun_out/final2/bench_full.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/final2/bench_full.json'))
print(round(d['value'],1), round(d['roofline']['frac'],4), d['config']['parity_spot_check'])
for k,v in d.get('secondary',{}).items(): print(' ', k, round(v['value'],3), v['unit'], round(v['roofline']['frac'],3), v['kernel'][:40], '|', v['parity'][:60])
PY
