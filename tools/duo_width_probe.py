"""Two-wave kernel (k_duo) or single-wave kernel (k_wave<16>) for biquad banks of 6144 .. 12 032 channels?  The dispatcher's
switch at 8192 lanes dates from round 2 (profiles/r02_bank_width_sweep.log), before k_duo had its storing wave and non-temporal
tiles.  -DALZ_TUNING build: ALZ_DUO_MAX_LANES moves the switch (read once per process).  2^30 samples per launch, median of 5."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  import torch
  import bench
  import audiolazy_amd as alz
  total = 1 << 30
  for lay in ("time", "chan"):
    for C in (6144, 7168, 8192, 9216, 10240, 12032):
      N = (total // C) // 64 * 64
      b, a = bench.resonator_coefs(C)
      shape = (N, C) if lay == "time" else (C, N)
      g = torch.Generator(device="cuda").manual_seed(3)
      x = torch.empty(shape, dtype=torch.float64, device="cuda").uniform_(-1.0, 1.0, generator=g)
      y = torch.empty_like(x)
      bank = alz.FilterBank([(b, a)], n_inputs=C, device=0)
      ms = []
      for _ in range(6):
        bank.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bank.process(x, layout=lay, out=y)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
      ms = sorted(ms[1:])
      print(json.dumps({"switch": os.environ.get("ALZ_DUO_MAX_LANES", "8192 (shipped)"), "layout": lay, "channels": C, "samples": N,
                        "kernel": bank.last_kernel, "gsamples_s": round(C * N / ms[2] / 1e6, 1)}), flush=True)
      del x, y


if __name__ == "__main__":
  main()
