#!/bin/bash
# LPC bit-exact kernel: products-then-additions (split2) and the full-chunk fast path (f1), per-call times; parity by bench
mkdir -p gpurun_out/r02lpc
for v in base lpc_split2 lpc_s0_f1 lpc_s2_f1 base; do
  if [ $v = base ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$PWD/tools/variants/$v.so; fi
  python - "$v" <<'PY'
import sys; sys.path.insert(0,'.')
import torch, time, numpy as np
from audiolazy_amd.lpc import kautocor_frames
from oracle import oracle
F, L, order = 65536, 480, 16
sig = torch.rand(F*L, dtype=torch.float64, device='cuda')*2-1
out=[]
for kw in [dict(), dict(exact=True)]:
  for _ in range(5): kautocor_frames(sig, L, order, **kw)
  torch.cuda.synchronize(); t=time.perf_counter()
  for _ in range(100): kautocor_frames(sig, L, order, **kw)
  torch.cuda.synchronize(); dt=(time.perf_counter()-t)/100
  out.append("%s %.1f us %.3f Gframes/s" % (kw, dt*1e6, F/dt/1e9))
c, e, st = kautocor_frames(sig, L, order, exact=True)
nf = 2048
rc, re, rs = oracle.kautocor_frames(sig[:nf*L].cpu().numpy(), nf, L, L, order)
ok = np.array_equal(c[:nf].cpu().numpy().view(np.uint64), rc.view(np.uint64))
print(sys.argv[1], " | ".join(out), "| exact mode bit-identical on %d frames: %s" % (nf, ok))
PY
done 2>/dev/null | tee gpurun_out/r02lpc/split.log
