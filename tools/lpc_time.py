import sys, time, ctypes
sys.path.insert(0, '.')
import torch, numpy as np
import audiolazy_amd as alz
from audiolazy_amd import _ffi
L = _ffi.load()
import os
F, N, order = int(os.environ.get("LPC_FRAMES", "65536")), 480, 16
sig = torch.rand(F * N, dtype=torch.float64, device='cuda') * 2 - 1
coefs = torch.empty((F, order + 1), dtype=torch.float64, device='cuda')
err = torch.empty(F, dtype=torch.float64, device='cuda'); st = torch.empty(F, dtype=torch.int32, device='cuda')
r = torch.empty((F, order + 1), dtype=torch.float64, device='cuda')
def t(fn, n=10):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
print("kautocor ms", t(lambda: L.alz_lpc_kautocor_dev(sig.data_ptr(), F, N, N, order, coefs.data_ptr(), err.data_ptr(), st.data_ptr(), 0, s)))
print("acorr    ms", t(lambda: L.alz_acorr_dev(sig.data_ptr(), F, N, N, order, r.data_ptr(), 0, s)))
print("levinson ms", t(lambda: L.alz_levinson_dev(r.data_ptr(), F, order + 1, order, coefs.data_ptr(), err.data_ptr(), st.data_ptr(), 0, s)))
