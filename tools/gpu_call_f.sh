#!/bin/bash
mkdir -p gpurun_out/r02f
export TMPDIR=/tmp
R=$(pwd)
ALZ_LIBRARY=$R/tools/variants/casc_dephase.so timeout 600 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_filters_api.py tests/test_gpu_bank.py -q -k "gammatone or cascade or casc or outer or filterbank" > gpurun_out/r02f/pytest_dephase.log 2>&1
tail -4 gpurun_out/r02f/pytest_dephase.log
g() { python bench.py --workload gammatone --steps 10 --warmup 2 $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gs/s %s %s' % (d['value'], d['config']['kernel'], d['config']['parity_spot_check']))"; }
echo "gammatone shipped (overlap 1): $(g)" | tee gpurun_out/r02f/gammatone.log
echo "gammatone dephase: $(ALZ_LIBRARY=$R/tools/variants/casc_dephase.so g)" | tee -a gpurun_out/r02f/gammatone.log
echo "gammatone shipped again: $(g)" | tee -a gpurun_out/r02f/gammatone.log
echo "gammatone dephase again: $(ALZ_LIBRARY=$R/tools/variants/casc_dephase.so g)" | tee -a gpurun_out/r02f/gammatone.log
f() { python bench.py --workload fir --steps 4 --warmup 1 $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gs/s %.1f TFLOP/s %s | %s' % (d['value'], d['roofline']['achieved'], d['config']['kernel'], d['config']['parity_spot_check'][:60]))"; }
echo "fir exact R48: $(ALZ_LIBRARY=$R/tools/variants/fir_R48_K8_W2.so f)" | tee gpurun_out/r02f/fir.log
echo "fir fma R48: $(ALZ_LIBRARY=$R/tools/variants/fir_R48_K8_W2.so f --fused)" | tee -a gpurun_out/r02f/fir.log
echo "fir fma R32W3: $(ALZ_LIBRARY=$R/tools/variants/fir_R32_K8_W3.so f --fused)" | tee -a gpurun_out/r02f/fir.log
