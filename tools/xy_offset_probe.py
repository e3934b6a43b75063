#!/usr/bin/env python3
"""configs[1] as channel-major rows [4096, 2^20]: does the kernel's time depend on WHERE the output block lies relative to the
input block?  One allocation, x at its start, y at x + 32 GiB + offset for a list of offsets; bit-exact and FMA mode.
(Round 6: the same kernel on the same shape read 12.3 ms in one process and 13.8 ms in another on one box.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import audiolazy_amd as alz
import bench

C, N = 4096, 1 << 20
layout = sys.argv[1] if len(sys.argv) > 1 else "chan"
offs = [int(v) for v in os.environ["XY_OFFS"].split(",")] if os.environ.get("XY_OFFS") else [0, 256, 2048, 8192, 16384, 1 << 20, 1 << 21, 3 << 20, (1 << 23) + 2048]
b, a = bench.resonator_coefs(C)
extra = (max(offs) // 8) + 1024
big = torch.empty(2 * C * N + extra, dtype=torch.float64, device="cuda")
shape = (C, N) if layout == "chan" else (N, C)
x = big[:C * N].view(shape)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x.copy_(torch.rand(shape, generator=g, device="cuda", dtype=torch.float64) * 2 - 1)
print("x at 0x%x" % x.data_ptr())
for fused in ((True,) if os.environ.get("XY_FMA_ONLY") else (False, True)):
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  if fused:
    bank.set_fused(True)
  for off in offs:
    y = big[C * N + off // 8: C * N + off // 8 + C * N].view(shape)
    bank.reset()
    for _ in range(2):
      bank.process(x, layout=layout, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
      bank.process(x, layout=layout, out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%s %-5s y - x - 32 GiB = %9d B: %.3f ms  %.1f Gsamples/s  (%s)" % (layout, "fma" if fused else "exact", off, ms, C * N / ms / 1e6, bank.last_kernel), flush=True)
