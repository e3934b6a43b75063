#!/bin/bash
# LPC: parity suite, the bench line, and per-call times of the three modes
mkdir -p gpurun_out/r02lpc
timeout 900 python -m pytest tests/test_gpu_lpc.py tests/test_gpu_fullwidth.py -x -q -m gpu > gpurun_out/r02lpc/pytest.log 2>&1
tail -3 gpurun_out/r02lpc/pytest.log
timeout 300 python bench.py --workload lpc --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tee gpurun_out/r02lpc/lpc.json | cut -c1-200
python - <<'PY' | tee gpurun_out/r02lpc/modes.log
import sys; sys.path.insert(0,'.')
import torch, time, numpy as np
from audiolazy_amd.lpc import kautocor_frames
sig = torch.rand(65536*480, dtype=torch.float64, device='cuda')*2-1
for kw in [dict(), dict(exact=True), dict(fused=True)]:
  for _ in range(3): kautocor_frames(sig, 480, 16, **kw)
  torch.cuda.synchronize(); t=time.perf_counter()
  for _ in range(50): kautocor_frames(sig, 480, 16, **kw)
  torch.cuda.synchronize(); dt=(time.perf_counter()-t)/50
  print(kw, "%.1f us per call, %.3f Gframes/s" % (dt*1e6, 65536/dt/1e9))
PY
