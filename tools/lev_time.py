"""The dense (bit-identical) Levinson-Durbin as a kernel of its own (k_levinson_dense<17>: lags from memory) on 65 536 frames of
order 16 -- the tail of k_acorr_stage<17, 2> in isolation.  usage: [ALZ_LIBRARY=...] python tools/lev_time.py"""
import ctypes, os, sys
sys.path.insert(0, '.')
import torch
from audiolazy_amd import _ffi
L = _ffi.load()
F, order = int(os.environ.get("LEV_FRAMES", "65536")), 16
g = torch.Generator(device="cuda"); g.manual_seed(3)
sig = torch.rand((F, 480), dtype=torch.float64, device="cuda", generator=g) * 2 - 1
r = torch.stack([(sig[:, :480 - k] * sig[:, k:]).sum(dim=1) for k in range(order + 1)], dim=1).contiguous()
coefs = torch.empty((F, order + 1), dtype=torch.float64, device="cuda")
err = torch.empty(F, dtype=torch.float64, device="cuda")
st = torch.empty(F, dtype=torch.int32, device="cuda")
def call():
  s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  _ffi.check(L.alz_levinson_dev_ex(r.data_ptr(), F, order + 1, order, coefs.data_ptr(), err.data_ptr(), st.data_ptr(), _ffi.LPC_DENSE, 0, s))
for _ in range(3): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): call()
e1.record(); torch.cuda.synchronize()
print("%s: dense Levinson-Durbin alone, %d frames: %.2f us per launch; checksum %.17g" % (os.environ.get("ALZ_LIBRARY", "shipped").split("/")[-1], F, e0.elapsed_time(e1) / 50 * 1e3, float(coefs.sum())))
