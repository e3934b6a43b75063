#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry (alz_bank_process_host): NumPy in, NumPy out."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import audiolazy_amd as alz

s, Hz = alz.sHz(48000)
filt = alz.resonator.z_exp(1000 * Hz, 100 * Hz)
for C, N in ((4096, 1 << 14), (4096, 1 << 16), (512, 1 << 18)):
  bank = alz.FilterBank([(filt.numlist, filt.denlist)], n_inputs=C)
  x = np.random.default_rng(0).uniform(-1, 1, (N, C))
  y = np.empty_like(x)
  bank.process(x, out=y)
  t0 = time.perf_counter()
  reps = 3
  for _ in range(reps):
    bank.process(x, out=y)
  dt = (time.perf_counter() - t0) / reps
  print("host path C=%d N=%d: %.1f ms  %.2f Gsamples/s  (%.1f GB/s over the link, both directions summed)"
        % (C, N, dt * 1e3, C * N / dt / 1e9, 16.0 * C * N / dt / 1e9))
