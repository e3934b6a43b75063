"""Run-to-wave mappings of k_fir_ring (csrc/alz_fir.hip, launch_fir) on configs[2]: 256 taps x 8192 channels x 2^18 rows.

Needs a -DALZ_TUNING build (ALZ_LIBRARY=tools/variants/libalzhip_tuning.so): the mapping and the pacing constants are
read per launch from the environment (the shipped library takes launch_fir's defaults: configuration "auto").  Every configuration's output is compared BITWISE with the shipped mapping's, then timed (median of 5
launches of ~35 ms between host synchronisations).  With --only NAME one configuration is launched three times and nothing else: the form
tools/r05_call2*.sh put under rocprofv3 --pmc FETCH_SIZE.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIGS = {
  "map1": {"ALZ_FIR_MAP": "1"},                                  # rounds 3 - 4: interleaved runs
  "auto": {},                                                    # what ships: launch_fir's choice of chain width and pacing
  "forced": {"ALZ_FIR_MAP": "4"},                                # chains by launch_fir's width rule even where its gate says no
  "free": {"ALZ_FIR_MAP": "4", "ALZ_FIR_PACED": "0"},            # the chains' mapping without the pacing
  "w4": {"ALZ_FIR_MAP": "4", "ALZ_FIR_W": "4", "ALZ_FIR_PACED": "1"},   # chains of W waves, paced
  "w8": {"ALZ_FIR_MAP": "4", "ALZ_FIR_W": "8", "ALZ_FIR_PACED": "1"},
  "w16": {"ALZ_FIR_MAP": "4", "ALZ_FIR_W": "16", "ALZ_FIR_PACED": "1"},
  "b50": {"ALZ_FIR_BOUND": "50"},                                # a wait gives up after this many percent of the lead
  "b150": {"ALZ_FIR_BOUND": "150"},
  "b200": {"ALZ_FIR_BOUND": "200"},
  "b300": {"ALZ_FIR_BOUND": "300"},
  "h100": {"ALZ_FIR_HANDOVER": "100"},
  "s100": {"ALZ_FIR_SHARE": "100"},
}
KEYS = ("ALZ_FIR_MAP", "ALZ_FIR_PACED", "ALZ_FIR_SHARE", "ALZ_FIR_HANDOVER", "ALZ_FIR_CFILL", "ALZ_FIR_W", "ALZ_FIR_BOUND")


def set_env(cfg, fused):
  for k in KEYS:
    os.environ.pop(k, None)
  os.environ.update(cfg)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--only")
  ap.add_argument("--configs", help="comma-separated subset for the timing pass")
  ap.add_argument("--fused", type=int, default=0)
  ap.add_argument("--channels", type=int, default=8192)
  ap.add_argument("--rows", type=int, default=1 << 18)
  ap.add_argument("--taps", type=int, default=256, help="keep the first TAPS taps (1: no halo -- every input byte is read once)")
  ap.add_argument("--pad", type=int, default=0, help="doubles between rows (the default pitch, 64 KiB, is a power of two)")
  args = ap.parse_args()
  import torch
  import bench
  import audiolazy_amd as alz
  dev = torch.device("cuda:0")
  C, N = args.channels, args.rows
  taps = bench.fir_taps(max(256, args.taps))[:args.taps]
  g = torch.Generator(device=dev).manual_seed(5)
  x = torch.empty((N, C + args.pad), dtype=torch.float64, device=dev).uniform_(-1.0, 1.0, generator=g)[:, :C]
  y = torch.empty((N, C + args.pad), dtype=torch.float64, device=dev)[:, :C]
  bank = alz.FilterBank([(taps, np.array([1.0]))], n_inputs=C, device=0)
  if args.fused:
    bank.set_fused(True)

  def run(cfg):
    set_env(cfg, args.fused)
    bank.reset()
    bank.process(x, layout="time", out=y)

  if args.only:
    for _ in range(3):
      run(CONFIGS[args.only])
    torch.cuda.synchronize()
    return
  run(CONFIGS["map1"])
  ref = y.clone()
  for name, cfg in CONFIGS.items():
    if args.configs and name not in args.configs.split(","):
      continue
    run(cfg)
    same = bool(torch.equal(ref.contiguous().view(torch.int64), y.contiguous().view(torch.int64)))
    ms = []
    for _ in range(5):
      bank.reset()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      bank.process(x, layout="time", out=y)
      torch.cuda.synchronize()
      ms.append((time.perf_counter() - t0) * 1e3)
    ms.sort()
    print(json.dumps({"config": name, "fused": args.fused, "kernel": bank.last_kernel, "bitwise_equal_to_map1": same,
                      "ms_median": round(ms[2], 3), "ms_min_max": [round(ms[0], 3), round(ms[-1], 3)],
                      "gsamples_s": round(C * N / ms[2] / 1e6, 2), "taps": len(taps), "pad": args.pad}), flush=True)


if __name__ == "__main__":
  main()
