#!/bin/bash
mkdir -p gpurun_out/r02l
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_filters_api.py -q -k "cfg4 or gammatone or cascade" > gpurun_out/r02l/pytest.log 2>&1
tail -4 gpurun_out/r02l/pytest.log
g() { python bench.py --workload gammatone --steps 10 --warmup 2 $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gs/s %s %s' % (d['value'], d['config']['kernel'], d['config']['parity_spot_check'][:100]))"; }
echo "gammatone exact: $(g)" | tee gpurun_out/r02l/gammatone.log
echo "gammatone fma: $(g --fused)" | tee -a gpurun_out/r02l/gammatone.log
echo "gammatone fma again: $(g --fused)" | tee -a gpurun_out/r02l/gammatone.log
