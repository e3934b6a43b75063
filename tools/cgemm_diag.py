"""Diagnostic for the experimental dot-product zero-state pass (tools/experiments/alz_scan_gemm.hip): the same block through the
variant library with ALZ_CSCAN_GEMM=0 and =1, differences per band and chunk."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import audiolazy_amd as alz
B, N = 256, 16384
s_, Hz = alz.sHz(48000)
fcs = [f * Hz for f in alz.erb_space(50., 20000., B)]
bank = alz.gammatone_bank(fcs, 1, strategy="slaney", Hz=Hz, device=0)
bank.set_time_parallel(True)
x = torch.rand((1, N), dtype=torch.float64, device="cuda") * 2 - 1
out = {}
for flag in ("0", "1"):
  os.environ["ALZ_CSCAN_GEMM"] = flag
  bank.reset()
  out[flag] = bank.process(x, layout="chan").cpu().numpy()
  print("flag", flag, "kernel", bank.last_kernel, "finite", np.isfinite(out[flag]).all())
a, b = out["0"], out["1"]
L = 256
scale = np.abs(a).max(axis=1, keepdims=True) + 1e-300
d = np.abs(a - b) / scale
print("max normalised diff %.3g" % d.max())
for band in (0, 1, 2, 3, 100, 254, 255):
  per_chunk = d[band].reshape(-1, L).max(axis=1)
  print("band %3d chunks 0..7: %s   worst chunk %d (%.2e)" % (band, " ".join("%.1e" % v for v in per_chunk[:8]), per_chunk.argmax(), per_chunk.max()))
j = 1
print("band 0, chunk 1, first samples old:", a[0, j * L:j * L + 4], "new:", b[0, j * L:j * L + 4])
print("band 1, chunk 1, first samples old:", a[1, j * L:j * L + 4], "new:", b[1, j * L:j * L + 4])
