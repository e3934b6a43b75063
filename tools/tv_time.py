#!/usr/bin/env python3
"""Time the time-varying filter kernel (alz_tv.hip): biquad with per-sample coefficient series."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiolazy_amd import timevar


def timed(fn, reps=5):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e-3


for C, N, layout in ((1, 1 << 20, "time"), (64, 1 << 18, "time"), (4096, 1 << 16, "time"), (4096, 1 << 16, "chan"),
                     (65536, 1 << 12, "time")):
  shape = (N, C) if layout == "time" else (C, N)
  x = torch.empty(shape, dtype=torch.float64, device="cuda").uniform_(-1, 1)
  b0 = torch.empty(N, dtype=torch.float64, device="cuda").uniform_(.1, 1)
  a1 = torch.empty(N, dtype=torch.float64, device="cuda").uniform_(-.9, .9)
  a2 = torch.empty(N, dtype=torch.float64, device="cuda").uniform_(-.3, .3)
  t = timed(lambda: timevar.process_block([b0, 0., -.5], [1., a1, a2], x, layout=layout))
  print("shared series   C=%6d N=%8d %s: %8.3f ms  %8.3f Gsamples/s" % (C, N, layout, t * 1e3, C * N / t / 1e9))
  if C > 1 and C * N <= (1 << 28):
    a1c = torch.empty(shape, dtype=torch.float64, device="cuda").uniform_(-.9, .9)
    t = timed(lambda: timevar.process_block([b0, 0., -.5], [1., a1c, a2], x, layout=layout))
    print("per-channel a1  C=%6d N=%8d %s: %8.3f ms  %8.3f Gsamples/s" % (C, N, layout, t * 1e3, C * N / t / 1e9))
