#!/bin/bash
mkdir -p gpurun_out/r02e
export TMPDIR=/tmp
R=$(pwd)
timeout 600 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_filters_api.py -q -k "gammatone or cascade or casc" > gpurun_out/r02e/pytest.log 2>&1
tail -4 gpurun_out/r02e/pytest.log
g() { python bench.py --workload gammatone --steps 10 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gs/s %s %s' % (d['value'], d['config']['kernel'], d['config']['parity_spot_check']))"; }
echo "gammatone default: $(g)" | tee gpurun_out/r02e/gammatone.log
echo "gammatone ALZ_PIPE_G=64: $(ALZ_PIPE_G=64 g)" | tee -a gpurun_out/r02e/gammatone.log
echo "gammatone default again: $(g)" | tee -a gpurun_out/r02e/gammatone.log
