"""k_tvpc (per-channel coefficient series): does it matter where the five arrays (x, b0, a1, a2, y) lie relative to
each other?  All are [N, C] float64 with N * C * 8 = 8 GiB at C = 4096, N = 2^18: torch hands out 2 MiB-aligned blocks,
so element (n, c) of every array has the same address modulo 2 MiB, and one tile step (64 rows x 32 KiB) is exactly
2 MiB.  Variants: arrays as allocated / staggered by k * STAG bytes / rows padded to ld = C + PAD.
usage: python tools/tvpc_alias.py"""
import ctypes
import sys

sys.path.insert(0, '.')
import torch
from audiolazy_amd import _ffi

L = _ffi.load()
C, N = 4096, 1 << 18


def run(stagger_bytes, pad_cols, label):
  ld = C + pad_cols
  per = N * ld + 1 << 0
  arrs = []
  for k in range(5):
    off = (k * stagger_bytes) // 8
    buf = torch.empty(N * ld + off + 16, dtype=torch.float64, device="cuda")
    view = buf[off:off + N * ld].view(N, ld)
    arrs.append((buf, view))
  x, b0, a1, a2, y = [v for _, v in arrs]
  x[:, :C].uniform_(-1, 1)
  b0[:, :C].uniform_(0.01, 0.02)
  a1[:, :C].uniform_(-1.5, -1.2)
  a2[:, :C].uniform_(0.5, 0.7)
  xh = torch.zeros((2, C), dtype=torch.float64, device="cuda")
  yh = torch.zeros((2, C), dtype=torch.float64, device="cuda")
  tb = (_ffi.TvTap * 3)(_ffi.TvTap(0.0, b0.data_ptr(), ld, 1, 0), _ffi.TvTap(0.0, None, 0, 0, 0), _ffi.TvTap(-0.015, None, 0, 0, 0))
  ta = (_ffi.TvTap * 3)(_ffi.TvTap(1.0, None, 0, 0, 0), _ffi.TvTap(0.0, a1.data_ptr(), ld, 1, 0), _ffi.TvTap(0.0, a2.data_ptr(), ld, 1, 0))
  s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

  def call():
    _ffi.check(L.alz_tv_process_dev(3, ctypes.cast(tb, ctypes.c_void_p), 3, ctypes.cast(ta, ctypes.c_void_p), C, x.data_ptr(),
                                    y.data_ptr(), N, _ffi.TIME_MAJOR, ld, ld, xh.data_ptr(), yh.data_ptr(), 0.0, 0, s))
  call()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(4):
    call()
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 4
  print("%-44s %8.3f ms  %7.1f Gsamples/s  %5.2f TB/s" % (label, ms, C * N / ms / 1e6, 40.0 * C * N / ms / 1e9))
  del arrs
  torch.cuda.empty_cache()


run(0, 0, "as allocated (2 MiB-aligned, ld = 4096)")
run(4096 + 256, 0, "staggered by k * 4352 B")
run(65536 + 4096 + 256, 0, "staggered by k * 69888 B")
run(2 * 1024 * 1024 // 5 // 256 * 256 + 256, 0, "staggered by k * ~410 KiB")
run(0, 16, "rows padded: ld = 4112")
run(4096 + 256, 16, "ld = 4112 and staggered by k * 4352 B")
