"""LPC timings (configs[4]): every mode of alz_lpc_kautocor_dev_ex at 65 536 and 1 Mi frames, as back-to-back
direct launches (bracketing events) and as a replayed HIP graph of one call.
usage: python tools/lpc_time.py   (LPC_FRAMES=65536,1048576)"""
import ctypes
import os
import sys

sys.path.insert(0, '.')
import torch
from audiolazy_amd import _ffi

L = _ffi.load()
N, order = 480, 16


def bracket(fn, n=40):
  fn(); fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3      # us


for F in [int(v) for v in os.environ.get("LPC_FRAMES", "65536,1048576").split(",")]:
  sig = torch.rand(F * N, dtype=torch.float64, device='cuda') * 2 - 1
  coefs = torch.empty((F, order + 1), dtype=torch.float64, device='cuda')
  err = torch.empty(F, dtype=torch.float64, device='cuda')
  st = torch.empty(F, dtype=torch.int32, device='cuda')
  r = torch.empty((F, order + 1), dtype=torch.float64, device='cuda')
  for name, flags in (("default (O(order^2) Levinson)", 0), ("dense = bit-identical", _ffi.LPC_DENSE), ("fma", _ffi.LPC_FUSED)):
    def call(stream=None):
      s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
      _ffi.check(L.alz_lpc_kautocor_dev_ex(sig.data_ptr(), F, N, N, order, coefs.data_ptr(), err.data_ptr(), st.data_ptr(),
                                           flags, 0, s))
    direct = bracket(call)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      call()
      with torch.cuda.graph(g, stream=side):
        call()
    torch.cuda.current_stream().wait_stream(side)
    graph = bracket(g.replay)
    g10 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
      with torch.cuda.graph(g10, stream=side):
        for _ in range(10):
          call()
    torch.cuda.current_stream().wait_stream(side)
    graph10 = bracket(g10.replay, 8) / 10
    print("F=%8d %-32s direct %8.2f us/call   graph(1 call) %8.2f   graph(10 calls) %8.2f per call   -> %.3f Gframes/s, %.1f %% of 8 TB/s"
          % (F, name, direct, graph, graph10, F / min(direct, graph, graph10) / 1e3, 3984.0 * F / min(direct, graph, graph10) / 1e6 / 8e6 * 100))
  s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  print("F=%8d acorr only %.2f us, dense Levinson only %.2f us" % (
      F, bracket(lambda: L.alz_acorr_dev(sig.data_ptr(), F, N, N, order, r.data_ptr(), 0, s)),
      bracket(lambda: L.alz_levinson_dev_ex(r.data_ptr(), F, order + 1, order, coefs.data_ptr(), err.data_ptr(), st.data_ptr(),
                                            _ffi.LPC_DENSE, 0, s))))
