"""The opt-in FMA mode of a biquad bank against the default kernel, by bank width (at configs[1]'s 4096
channels the FMA mode is SLOWER than the bit-exact kernel -- where does it help?).  Resonator bank of bench.py, time-major and
channel-major blocks of 2^30 samples in all, median of 5 launches, same process."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  import torch
  import bench
  import audiolazy_amd as alz
  total = 1 << 30
  for lay in ("time", "chan"):
    for C in (256, 512, 1024, 2048, 4096, 8192, 16384, 65536):
      N = total // C
      b, a = bench.resonator_coefs(C)
      shape = (N, C) if lay == "time" else (C, N)
      g = torch.Generator(device="cuda").manual_seed(3)
      x = torch.empty(shape, dtype=torch.float64, device="cuda").uniform_(-1.0, 1.0, generator=g)
      y = torch.empty_like(x)
      row = {"layout": lay, "channels": C, "samples": N}
      for fused in (False, True):
        bank = alz.FilterBank([(b, a)], n_inputs=C, device=0)
        bank.set_fused(fused)
        ms = []
        for _ in range(6):
          bank.reset()
          torch.cuda.synchronize()
          t0 = time.perf_counter()
          bank.process(x, layout=lay, out=y)
          torch.cuda.synchronize()
          ms.append((time.perf_counter() - t0) * 1e3)
        ms = sorted(ms[1:])
        row["fma" if fused else "default"] = {"kernel": bank.last_kernel, "gsamples_s": round(total / ms[2] / 1e6, 1)}
      row["fma_over_default"] = round(row["fma"]["gsamples_s"] / row["default"]["gsamples_s"], 3)
      print(json.dumps(row), flush=True)
      del x, y


if __name__ == "__main__":
  main()
