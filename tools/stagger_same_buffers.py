#!/usr/bin/env python3
"""The staggered start against the lock-step start ON THE SAME BUFFERS (the tuning variant of alz_wave.hip reads ALZ_DUO_STAGGER at
every launch): channel-major configs[1], FMA mode and bit-exact, several allocations per process.  Separates the stagger's effect from
the placement's (profiles/NOTES_r06.md 8.7)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiolazy_amd as alz
import bench

C, N = 4096, 1 << 20
b, a = bench.resonator_coefs(C)
banks = {"fma": alz.FilterBank([(b, a)], n_inputs=C).set_fused(True), "exact": alz.FilterBank([(b, a)], n_inputs=C)}

def timed(bank, x, y):
  bank.reset()
  bank.process(x, layout="chan", out=y)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(4):
    bank.process(x, layout="chan", out=y)
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / 4

for rnd in range(4):
  one = rnd % 2 == 1
  if one:
    big = torch.empty(2 * C * N, dtype=torch.float64, device="cuda")
    x, y = big[:C * N].view(C, N), big[C * N:].view(C, N)
  else:
    x = torch.empty((C, N), dtype=torch.float64, device="cuda")
    y = torch.empty((C, N), dtype=torch.float64, device="cuda")
  x.uniform_(-1, 1)
  for mode, bank in banks.items():
    row = []
    for st in (0, 20, 80, 0, 20, 80):
      os.environ["ALZ_DUO_STAGGER"] = str(st)
      row.append("%d:%.2f" % (st, timed(bank, x, y)))
    print("alloc %d (%s) %-5s stagger:ms  %s" % (rnd, "one allocation" if one else "two allocations", mode, "  ".join(row)), flush=True)
  del x, y
  if one:
    del big
  torch.cuda.empty_cache()
