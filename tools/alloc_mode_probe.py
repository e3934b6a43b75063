#!/usr/bin/env python3
"""channel-major configs[1], FMA mode (shipped library: staggered start): x and y as two allocations against x and y carved from
one allocation, in the same process, alternating.  (profiles/NOTES_r06.md 8.7: is the slow mode a property of the allocation?)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiolazy_amd as alz
import bench

C, N = 4096, 1 << 20
b, a = bench.resonator_coefs(C)
bank = alz.FilterBank([(b, a)], n_inputs=C).set_fused(True)

def timed(x, y):
  bank.reset()
  for _ in range(2):
    bank.process(x, layout="chan", out=y)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(5):
    bank.process(x, layout="chan", out=y)
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / 5

for rnd in range(3):
  x = torch.empty((C, N), dtype=torch.float64, device="cuda").uniform_(-1, 1)
  y = torch.empty((C, N), dtype=torch.float64, device="cuda")
  t_sep = timed(x, y)
  d_sep = y.data_ptr() - x.data_ptr()
  del x, y
  torch.cuda.empty_cache()
  big = torch.empty(2 * C * N + 2048, dtype=torch.float64, device="cuda")
  x = big[:C * N].view(C, N)
  x.uniform_(-1, 1)
  t_one = [timed(x, big[C * N + o: 2 * C * N + o].view(C, N)) for o in (0, 1024)]
  del x, big
  torch.cuda.empty_cache()
  print("round %d: two allocations (y - x = %d MiB + %d B): %.2f ms; one allocation, y - x = 32 GiB / + 8 KiB: %.2f / %.2f ms"
        % (rnd, d_sep >> 20, d_sep & ((1 << 20) - 1), t_sep, t_one[0], t_one[1]), flush=True)
