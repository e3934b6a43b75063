#!/bin/bash
# round-2 GPU call D: k_pipe<32> parity + timing, FIR FMA register-tile sweep, time-parallel kernel trace
mkdir -p gpurun_out/r02d
export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_scan.py tests/test_gpu_filters_api.py tests/test_gpu_reference_tests2.py -q > gpurun_out/r02d/pytest.log 2>&1
tail -12 gpurun_out/r02d/pytest.log
g() { python bench.py --workload gammatone --steps 10 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gs/s %s %s' % (d['value'], d['config']['kernel'], d['config']['parity_spot_check']))"; }
echo "gammatone default: $(g)" | tee gpurun_out/r02d/gammatone.log
echo "gammatone ALZ_PIPE_G=64: $(ALZ_PIPE_G=64 g)" | tee -a gpurun_out/r02d/gammatone.log
echo "gammatone default again: $(g)" | tee -a gpurun_out/r02d/gammatone.log
f() { python bench.py --workload fir --steps 4 --warmup 1 $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gs/s %.1f TFLOP/s %s | %s' % (d['value'], d['roofline']['achieved'], d['config']['kernel'], d['config']['parity_spot_check'][:60]))"; }
echo "fir fma shipped (R32 K8 W2): $(f --fused)" | tee gpurun_out/r02d/fir_sweep.log
for v in tools/variants/fir_*.so; do
  echo "fir fma $(basename $v): $(ALZ_LIBRARY=$R/$v f --fused)" | tee -a gpurun_out/r02d/fir_sweep.log
done
echo "fir exact shipped: $(f)" | tee -a gpurun_out/r02d/fir_sweep.log
for v in tools/variants/fir_R32_K8_W3.so tools/variants/fir_R24_K8_W3.so tools/variants/fir_R16_K8_W4.so; do
  echo "fir exact $(basename $v): $(ALZ_LIBRARY=$R/$v f)" | tee -a gpurun_out/r02d/fir_sweep.log
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02d/trace_tp -o t -- python $R/bench.py --channels 512 --time-parallel 1 --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-parity-check > $R/gpurun_out/r02d/trace_tp.log 2>&1
cd $R
cat gpurun_out/r02d/trace_tp/*kernel_stats.csv 2>/dev/null | head -12
tail -2 gpurun_out/r02d/trace_tp.log
