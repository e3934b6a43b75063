#!/bin/bash
# LPC stage-ring depth: per-call times of the three modes, shipped (4 slots) against 3 and 5; bench parity check
mkdir -p gpurun_out/r02lpc
for v in base lpc_ring3 lpc_ring5 base; do
  if [ $v = base ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$PWD/tools/variants/$v.so; fi
  python - "$v" <<'PY'
import sys; sys.path.insert(0,'.')
import torch, time, numpy as np
from audiolazy_amd.lpc import kautocor_frames
sig = torch.rand(65536*480, dtype=torch.float64, device='cuda')*2-1
out=[]
for kw in [dict(), dict(fused=True)]:
  for _ in range(5): kautocor_frames(sig, 480, 16, **kw)
  torch.cuda.synchronize(); t=time.perf_counter()
  for _ in range(100): kautocor_frames(sig, 480, 16, **kw)
  torch.cuda.synchronize(); dt=(time.perf_counter()-t)/100
  out.append("%s %.1f us %.3f Gframes/s" % (kw, dt*1e6, 65536/dt/1e9))
print(sys.argv[1], " | ".join(out))
PY
done 2>/dev/null | tee gpurun_out/r02lpc/ring.log
unset ALZ_LIBRARY
timeout 300 python bench.py --workload lpc --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | cut -c1-120,380-900
timeout 300 python bench.py --workload lpc --fused --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | cut -c1-120,380-900
