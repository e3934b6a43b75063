"""One stream x 256 gammatone bands x 2^20 in the time-parallel mode: chunk length against throughput, both layouts
(auto = 4096: 256 chunks = 1024 cascade groups, one wave per SIMD in the replay; 2048 -> two waves per SIMD but twice the
serial chunk recursion; 1024 -> four).   python tools/tp_chunk_sweep.py"""
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch
import audiolazy_amd as alz

s_, Hz = alz.sHz(48000)
fcs = [f * Hz for f in alz.erb_space(50., 20000., 256)]
N = 1 << 20
g = torch.Generator(device='cuda').manual_seed(5)
for layout in ("chan", "time"):
  x = torch.empty((1, N) if layout == "chan" else (N, 1), dtype=torch.float64, device='cuda').uniform_(-1, 1, generator=g)
  y = torch.empty((256, N) if layout == "chan" else (N, 256), dtype=torch.float64, device='cuda')
  for chunk in (True, 2048, 1024, 8192):
    bank = alz.gammatone_bank(fcs, 1, strategy="slaney", Hz=Hz).set_time_parallel(chunk)
    bank.reset()
    for _ in range(5):
      bank.process(x, layout=layout, out=y)
    best = []
    for rep in range(3):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(40):
        bank.process(x, layout=layout, out=y)
      torch.cuda.synchronize()
      best.append((time.perf_counter() - t0) / 40)
    ms = sorted(best)[1] * 1e3
    print("layout %-4s chunk %-5s %-34s %.4f ms  %.1f Gsamples/s" % (layout, "auto" if chunk is True else chunk, bank.last_kernel, ms, 256 * N / ms / 1e6))
