"""k_fir_ring's paced chains when the chip is shared: two FIR banks (configs[2] halves: 256 taps x 8192 channels x 2^17 rows
each) launched on two streams at once, against the same two launches one after the other -- with the shipped mapping and,
in a -DALZ_TUNING build, with the interleaved mapping of rounds 3 - 4 (ALZ_FIR_MAP=1).  A chain's waves pace each other by
start stamps and give a wait up after 1.5 leads: with another kernel's blocks in the way the stamps come late or not at
all, and what that may cost is bounded by those waits.  Outputs are compared bitwise with a launch that had the chip alone."""
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
  C, N = 8192, 1 << 17
  taps = bench.fir_taps()
  g = torch.Generator(device="cuda").manual_seed(9)
  xs = [torch.empty((N, C), dtype=torch.float64, device="cuda").uniform_(-1.0, 1.0, generator=g) for _ in range(2)]
  ys = [torch.empty_like(x) for x in xs]
  banks = [alz.FilterBank([(taps, np.array([1.0]))], n_inputs=C, device=0) for _ in range(2)]
  streams = [torch.cuda.Stream(), torch.cuda.Stream()]
  fused = len(sys.argv) > 1 and sys.argv[1] == "fma"
  for b in banks:
    if fused:
      b.set_fused(True)
  for mapping in (["shipped", "interleaved"] if os.environ.get("ALZ_LIBRARY") else ["shipped"]):
    if mapping == "interleaved":
      os.environ["ALZ_FIR_MAP"] = "1"
    else:
      os.environ.pop("ALZ_FIR_MAP", None)
    alone = []
    for i in range(2):
      banks[i].reset()
      alone.append(banks[i].process(xs[i], layout="time").clone())
    torch.cuda.synchronize()
    res = {}
    for mode in ("one after the other", "two streams"):
      ms = []
      for _ in range(5):
        for b in banks:
          b.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2):
          if mode == "two streams":
            with torch.cuda.stream(streams[i]):
              banks[i].process(xs[i], layout="time", out=ys[i])
          else:
            banks[i].process(xs[i], layout="time", out=ys[i])
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
      ms.sort()
      same = all(bool(torch.equal(ys[i].view(torch.int64), alone[i].view(torch.int64))) for i in range(2))
      res[mode] = {"ms_median": round(ms[2], 3), "ms_min_max": [round(ms[0], 3), round(ms[-1], 3)], "bitwise_equal_to_alone": same}
    print(json.dumps({"mapping": mapping, "fused": fused, "kernel": banks[0].last_kernel, **res}), flush=True)


if __name__ == "__main__":
  main()
