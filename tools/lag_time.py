"""lag_matrix for a batch of frames (alz_lag_matrix_dev): the row-per-lane kernel (curated orders) against the cell-per-lane one.
Usage: python tools/lag_time.py [frames] [frame_len]"""
import ctypes, sys
sys.path.insert(0, '.')
import torch
from audiolazy_amd import _ffi
F = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
Ln = int(sys.argv[2]) if len(sys.argv) > 2 else 480
L = _ffi.load()
sig = torch.rand(F * Ln, dtype=torch.float64, device="cuda") * 2 - 1
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for order in (16, 15, 8, 32):
  P = order + 1
  phi = torch.empty((F, P, P), dtype=torch.float64, device="cuda")
  run = lambda: _ffi.check(L.alz_lag_matrix_dev(sig.data_ptr(), F, Ln, Ln, order, phi.data_ptr(), 0, st))
  run(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(3): run()
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 3
  flop = 2.0 * F * P * P * (Ln - order)
  print("order %2d  %-20s %8.3f ms  %6.2f TFLOP/s (mul + add counted)  %6.3f Gframes/s" % (order, _ffi.last_kernel(), ms, flop / ms / 1e9, F / ms / 1e6))
