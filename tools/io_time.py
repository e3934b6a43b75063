#!/usr/bin/env python3
"""Time the streaming kernels either side of the path (alz_io.hip): PCM decode / encode and the
ordered mixdown, data resident in HBM, torch events on the launch stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiolazy_amd as alz
from audiolazy_amd.bank import mix_sets


def timed(fn, reps=10):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e-3


n = 1 << 28
for bits in (8, 16, 24, 32):
  raw = torch.randint(0, 256, (n * bits // 8,), dtype=torch.uint8, device="cuda")
  t = timed(lambda: alz.decode_pcm(raw, bits))
  print("decode %2d-bit: %.2f ms  %.1f Gsamples/s  %.0f GB/s (in+out)" % (bits, t * 1e3, n / t / 1e9, n * (bits / 8 + 8) / t / 1e9))
  del raw
x = torch.empty(n, dtype=torch.float64, device="cuda").uniform_(-1, 1)
for dfmt, w in (("f", 4), ("d", 8)):
  t = timed(lambda: alz.encode_pcm(x, dfmt))
  print("encode %s: %.2f ms  %.1f Gsamples/s  %.0f GB/s" % (dfmt, t * 1e3, n / t / 1e9, n * (8 + w) / t / 1e9))
xi = (x * 32767).round()
t = timed(lambda: alz.encode_pcm(xi, "h"))
print("encode h: %.2f ms  %.1f Gsamples/s  %.0f GB/s" % (t * 1e3, n / t / 1e9, n * 10 / t / 1e9))
t = timed(lambda: alz.encode_pcm(xi, "h", ">"))
print("encode >h (generic path): %.2f ms  %.1f Gsamples/s" % (t * 1e3, n / t / 1e9))
del x, xi
for layout, (B, S, N) in (("chan", (256, 64, 1 << 14)), ("time", (256, 64, 1 << 14)), ("time", (8, 4096, 1 << 13))):
  shape = (B * S, N) if layout == "chan" else (N, B * S)
  y = torch.empty(shape, dtype=torch.float64, device="cuda").uniform_(-1, 1)
  out = torch.empty((S, N) if layout == "chan" else (N, S), dtype=torch.float64, device="cuda")
  t = timed(lambda: mix_sets(y, B, S, layout=layout, out=out))
  print("mix %s %dx%dx%d: %.2f ms  %.0f GB/s read" % (layout, B, S, N, t * 1e3, B * S * N * 8 / t / 1e9))
  del y
