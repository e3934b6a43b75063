#!/bin/bash
mkdir -p gpurun_out/r02p
export TMPDIR=/tmp
for l in 14 15 16 17 20; do
  r=$(python bench.py --channels 512 --log2-samples $l --time-parallel 1 --steps 40 --warmup 4 --no-secondary --no-cpu-baseline --no-parity-check 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gs/s  %.3f ms/step  kernel-event %.3f ms  %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['config']['kernel']))")
  echo "512 ch x 2^$l samples per call, time-parallel: $r" | tee -a gpurun_out/r02p/mall.log
done
