#!/usr/bin/env python3
"""Differential fuzz of the HIP engine against the oracle: random bank structures (tap counts,
zero patterns, shared / per-channel coefficients, gains, 1-4 cascaded sections, DIAGONAL banks of
1-700 channels, both layouts, ragged block splits, memory / zero), bit-exact or it prints the
failing recipe.  Test infrastructure like the oracle itself; run on the GPU box:
    python tools/fuzz_bank.py [cases] [seed]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import audiolazy_amd as alz
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
bad = 0
kernels = {}
for case in range(cases):
  C = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 64, 65, 127, 200, 256, 300, 513, 700]))
  nsec = int(rng.choice([1, 1, 1, 2, 3, 4]))
  per_channel = bool(rng.integers(0, 2))
  N = int(rng.choice([1, 2, 5, 63, 64, 65, 200, 640, 1000, 4099]))
  layout = str(rng.choice(["time", "chan"]))
  secs, nbs, nas, bs, as_ = [], [], [], [], []
  for s in range(nsec):
    nb, na = int(rng.integers(1, 7)), int(rng.integers(1, 6))
    shape = (C,) if per_channel else ()
    b = rng.uniform(-1, 1, shape + (nb,))
    a = rng.uniform(-.3, .3, shape + (na,))
    a[..., 0] = rng.choice([1., 1., 1., 2., -1., .5])
    zb = rng.random(nb) < .3                         # uniform zero pattern, sometimes per channel
    za = rng.random(na) < .3
    za[0] = False
    b[..., zb] = 0.
    a[..., za] = 0.
    if per_channel and rng.random() < .3:            # ragged zero pattern
      b[rng.integers(0, C), rng.integers(0, nb)] = 0.
    secs.append((b, a)); nbs.append(nb); nas.append(na)
    bs.append(b.reshape(-1, nb) if per_channel else b); as_.append(a.reshape(-1, na) if per_channel else a)
  zero = float(rng.choice([0., 0., .25, -1.5]))
  bank = alz.FilterBank(secs, n_inputs=C)
  bank.reset(zero=zero)
  x = rng.uniform(-1, 1, (N, C) if layout == "time" else (C, N))
  bcat = np.concatenate(bs, axis=-1)
  acat = np.concatenate(as_, axis=-1)
  ref = oracle.bank(nbs, nas, bcat, acat, x, layout=layout, zero=zero)
  cut = int(rng.integers(0, N + 1))
  parts = []
  for lo, hi in ((0, cut), (cut, N)):
    blk = x[lo:hi] if layout == "time" else x[:, lo:hi]
    if blk.size:
      parts.append(bank.process(np.ascontiguousarray(blk), layout=layout))
  got = np.concatenate(parts, axis=0 if layout == "time" else 1)
  for name in set(bank.last_kernel.split("+")):
    kernels[name] = kernels.get(name, 0) + 1
  # bit for bit, except that a NaN is a NaN (an unstable random recipe ends in inf - inf; x86 and
  # gfx950 then produce quiet NaNs of opposite sign bit, which Python cannot tell apart)
  same = (got.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(got) & np.isnan(ref))
  if not same.all():
    bad += 1
    print("MISMATCH case %d: C=%d nsec=%d per_channel=%s N=%d cut=%d layout=%s nb=%s na=%s zero=%s kernel=%s maxdiff=%g"
          % (case, C, nsec, per_channel, N, cut, layout, nbs, nas, zero, bank.last_kernel, np.nanmax(np.abs(got - ref))))
    diff = np.argwhere(~same)
    ch_axis = 1 if layout == "time" else 0
    for ch in sorted(set(diff[:, ch_axis].tolist()))[:3]:
      where = diff[diff[:, ch_axis] == ch][:4]
      print("  channel %d: b=%s a=%s" % (ch, [s_[0][ch].tolist() if per_channel else s_[0].tolist() for s_ in secs],
                                       [s_[1][ch].tolist() if per_channel else s_[1].tolist() for s_ in secs]))
      for idx in where:
        print("    at %s got %r ref %r" % (idx.tolist(), got[tuple(idx)], ref[tuple(idx)]))
print("%d cases, %d mismatches; kernels: %s" % (cases, bad, kernels))
sys.exit(1 if bad else 0)
