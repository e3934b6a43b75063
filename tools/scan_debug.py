"""Where does the time-parallel result differ from the oracle?  Error per chunk, per configuration."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiolazy_amd as alz
from oracle import oracle
import bench

def run(C, N, chunk, layout="time"):
  b, a = bench.resonator_coefs(4096)
  st = 4096 // C
  b, a = b[::st].copy(), a[::st].copy()
  x = np.random.default_rng(1).uniform(-1, 1, (N, C))
  ref = oracle.bank([3], [3], b, a, x)
  bank = alz.FilterBank([(b, a)], n_inputs=C).set_time_parallel(chunk)
  bank.reset()
  y = bank.process(torch.from_numpy(x).cuda()).cpu().numpy()
  L = chunk if chunk is not True else max(256, (N // ((65536 + C - 1) // C)) // 64 * 64)
  K = N // L
  err = np.abs(y - ref)
  per_chunk = [float(err[j * L:(j + 1) * L].max()) for j in range(min(K, 6))]
  head = [float(err[j * L:j * L + 4].max()) for j in range(min(K, 6))]
  bad_ch = np.nonzero(err.max(axis=0) > 1e-9)[0]
  print("C=%d N=%d chunk=%s L=%d K=%d kernel=%s max=%.3g per-chunk=%s head=%s bad-channels=%d first=%s"
        % (C, N, chunk, L, K, bank.last_kernel, err.max(), ["%.2g" % v for v in per_chunk],
           ["%.2g" % v for v in head], len(bad_ch), bad_ch[:8]))

for C, N, chunk in [(512, 1 << 16, True), (512, 1 << 16, 2048), (128, 1 << 16, 64 * 4), (64, 1 << 16, 64), (1024, 1 << 14, 256)]:
  run(C, N, chunk)
