#!/bin/bash
mkdir -p gpurun_out/r02g
export TMPDIR=/tmp
R=$(pwd)
ALZ_PIPE=3 timeout 600 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_filters_api.py tests/test_gpu_bank.py tests/test_gpu_reference_tests2.py -q -k "gammatone or cascade or casc or outer or filterbank or auditory" > gpurun_out/r02g/pytest_tandem.log 2>&1
tail -6 gpurun_out/r02g/pytest_tandem.log
g() { timeout 120 python bench.py --workload gammatone --steps 10 --warmup 2 $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gs/s %s %s' % (d['value'], d['config']['kernel'], d['config']['parity_spot_check']))"; }
echo "gammatone shipped: $(g)" | tee gpurun_out/r02g/gammatone.log
echo "gammatone tandem: $(ALZ_PIPE=3 g)" | tee -a gpurun_out/r02g/gammatone.log
echo "gammatone tandem again: $(ALZ_PIPE=3 g)" | tee -a gpurun_out/r02g/gammatone.log
