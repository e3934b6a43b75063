"""Does a power-of-two row pitch cost HBM bandwidth?  The BASELINE shapes are all powers of two (4096 / 65536 channels x
2^16 .. 2^20 samples), so consecutive rows of a block lie 2^k bytes apart.  Same kernels, same work, leading dimensions
padded by a few hundred bytes (the C ABI takes any ld >= the row length):
  * wide biquad banks (k_wave<64>): time-major [N, C] with ld = C + pad, channel-major [C, N] with ld = N + pad;
  * the headline bank (k_duo) and the one-stream gammatone bank in time-parallel mode ([bands, N] rows 8 MiB apart).
    python tools/pitch_probe.py"""
import ctypes
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch
import audiolazy_amd as alz
from audiolazy_amd import _ffi
import bench

L = _ffi.load()


def run(bank, n, lay, cin, cout, pad_in, pad_out, steps=12):
  tm = lay == _ffi.TIME_MAJOR
  ldx = (cin if tm else n) + pad_in
  ldy = (cout if tm else n) + pad_out
  rows_x, rows_y = (n, n) if tm else (cin, cout)
  x = torch.empty(rows_x * ldx, dtype=torch.float64, device='cuda').uniform_(-1, 1)
  y = torch.empty(rows_y * ldy, dtype=torch.float64, device='cuda')
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  call = lambda: _ffi.check(L.alz_bank_process_dev(bank._h, x.data_ptr(), y.data_ptr(), n, lay, ldx, ldy, stream))
  for _ in range(3):
    call()
  best = []
  for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      call()
    torch.cuda.synchronize()
    best.append((time.perf_counter() - t0) / steps)
  del x, y
  torch.cuda.empty_cache()
  return sorted(best)[1], bank.last_kernel


for C, log2n in ((65536, 16), (16384, 18), (4096, 20)):
  n = 1 << log2n
  b, a = bench.resonator_coefs(C)
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  for lay, name in ((_ffi.TIME_MAJOR, "time-major"), (_ffi.CHAN_MAJOR, "channel-major")):
    for pad in (0, 32, 64, 272):
      bank.reset()
      sec, kern = run(bank, n, lay, C, C, pad, pad)
      print("biquad bank %6d ch x 2^%d %-13s pad %3d doubles: %8.3f ms  %6.1f Gsamples/s  %s" % (C, log2n, name, pad, sec * 1e3, C * n / sec / 1e9, kern))
  del bank

s_, Hz = alz.sHz(48000)
fcs = [f * Hz for f in alz.erb_space(50., 20000., 256)]
n = 1 << 20
for lay, name in ((_ffi.CHAN_MAJOR, "channel-major"), (_ffi.TIME_MAJOR, "time-major")):
  for pad in (0, 32, 272):
    bank = alz.gammatone_bank(fcs, 1, strategy="slaney", Hz=Hz).set_time_parallel(True)
    bank.reset()
    sec, kern = run(bank, n, lay, 1, 256, 0, pad, steps=40)
    print("one stream x 256 bands x 2^20, time-parallel, %-13s output pad %3d doubles: %7.4f ms  %6.1f Gsamples/s  %s" % (name, pad, sec * 1e3, 256 * n / sec / 1e9, kern))
# configs[3]: 256 bands x 64 streams x 2^16, channel-major rows 512 KiB apart
n = 1 << 16
for pad in (0, 32, 272):
  bank = alz.gammatone_bank(fcs, 64, strategy="slaney", Hz=Hz)
  bank.reset()
  sec, kern = run(bank, n, _ffi.CHAN_MAJOR, 64, 256 * 64, pad, pad, steps=20)
  print("configs[3] 256 bands x 64 streams x 2^16 channel-major, pad %3d doubles: %7.4f ms  %6.1f Gsamples/s  %s" % (pad, sec * 1e3, 256 * 64 * n / sec / 1e9, kern))
