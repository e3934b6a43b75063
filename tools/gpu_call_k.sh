#!/bin/bash
mkdir -p gpurun_out/r02k
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lpc.py tests/test_gpu_fullwidth.py tests/test_gpu_reference_tests2.py tests/test_gpu_bench_contract.py -q -k "lpc or acorr or kautocor or levinson or cfg5 or fused" > gpurun_out/r02k/pytest.log 2>&1
tail -5 gpurun_out/r02k/pytest.log
l() { python bench.py --workload lpc --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f Gframes/s kernel %.1f us frac %.3f | %s' % (d['value'], d['roofline']['kernel_ms_avg']*1e3, d['roofline']['frac'], d['config']['parity_spot_check'][:70]))"; }
echo "lpc paired waves: $(l)" | tee gpurun_out/r02k/lpc.log
echo "lpc ALZ_LPC_PAIR=0: $(ALZ_LPC_PAIR=0 l)" | tee -a gpurun_out/r02k/lpc.log
echo "lpc paired waves again: $(l)" | tee -a gpurun_out/r02k/lpc.log
python tools/lpc_time.py 2>/dev/null | tee -a gpurun_out/r02k/lpc.log
lf() { python bench.py --workload lpc --fused --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f Gframes/s kernel %.1f us frac %.3f | %s' % (d['value'], d['roofline']['kernel_ms_avg']*1e3, d['roofline']['frac'], d['config']['parity_spot_check'][:90]))"; }
echo "lpc FMA: $(lf)" | tee -a gpurun_out/r02k/lpc.log
echo "lpc FMA again: $(lf)" | tee -a gpurun_out/r02k/lpc.log
