#!/usr/bin/env python3
"""Differential fuzz of the STREAMING kernels (k_duo / k_wave / k_casc / k_pipe / k_fir / k_sparse)
against the oracle: curated tap patterns, bank sizes around the kernels' tile edges, both
layouts, DIAGONAL and OUTER banks, ragged block splits.  Bit-exact or it prints the recipe.
    python tools/fuzz_stream.py [cases] [seed]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import audiolazy_amd as alz
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
PATTERNS = [((1,), (1, 1)), ((1, 1), (1, 1)), ((1,), (1, 1, 1)), ((1, 1), (1, 1, 1)), ((1, 0, 1), (1, 1, 1)),
            ((1, 1, 1), (1, 1, 1)), ((1,), (1, 0, 1)), ((1, 1, 1), (1,)), ((1, 1), (1,))]


def section(C, per_channel, pattern):
  pb, pa = pattern
  shape = (C,) if per_channel else ()
  b = rng.uniform(-1, 1, shape + (len(pb),)) * np.array(pb, dtype=float)
  a = rng.uniform(-.45, .45, shape + (len(pa),)) * np.array(pa, dtype=float)
  a[..., 0] = 1.
  return b, a


bad, kernels = 0, {}
for case in range(cases):
  kind = str(rng.choice(["biquad", "biquad", "cascade", "fir", "comb", "outer"]))
  layout = str(rng.choice(["time", "chan"]))
  per_channel = bool(rng.integers(0, 2))
  C = int(rng.choice([16, 32, 48, 64, 80, 128, 256, 272, 1024, 4096, 4112]))
  N = int(rng.choice([64, 128, 129, 191, 192, 1000, 4096, 5000]))
  n_inputs, mode, n_sets = C, "diagonal", (C if per_channel else 1)
  if kind == "biquad":
    secs = [section(C, per_channel, PATTERNS[int(rng.integers(0, len(PATTERNS)))])]
  elif kind == "cascade":
    pats = [PATTERNS[int(rng.integers(0, 6))] for _ in range(int(rng.choice([2, 4])))]
    if rng.random() < .5:
      pats = [((1, 1), (1, 1, 1))] * 4                 # the gammatone.slaney shape
    secs = [section(C, per_channel, pt) for pt in pats]
  elif kind == "fir":
    nt = int(rng.choice([17, 40, 256]))
    C = min(C, 1024)
    n_inputs, n_sets = C, (C if per_channel else 1)
    secs = [(rng.uniform(-1, 1, ((C,) if per_channel else ()) + (nt,)), np.ones(((C,) if per_channel else ()) + (1,)))]
  elif kind == "comb":
    D = int(rng.choice([16, 37, 200]))
    C = min(C, 256)
    n_inputs, n_sets, per_channel = C, 1, False
    b, a = np.zeros(D + 1), np.zeros(D + 1)
    b[0], a[0], a[D] = 1., 1., -float(rng.uniform(.1, .9))
    if rng.random() < .5:
      b[D] = float(rng.uniform(-1, 1))
    secs = [(b, a)]
    layout = "time"
  else:                                                # OUTER: B coefficient sets on S inputs
    B, S = int(rng.choice([4, 16, 64])), int(rng.choice([16, 64]))
    C, n_inputs, mode, n_sets, per_channel = B * S, S, "outer", B, True
    pats = [((1, 1), (1, 1, 1))] * 4 if rng.random() < .5 else [PATTERNS[int(rng.integers(0, 6))]]
    secs = [section(B, True, pt) for pt in pats]
  zero = float(rng.choice([0., 0., .25]))
  bank = alz.FilterBank(secs, n_inputs=n_inputs, mode=mode)
  bank.reset(zero=zero)
  x = rng.uniform(-1, 1, (N, n_inputs) if layout == "time" else (n_inputs, N))
  # oracle: every channel with its own coefficient row
  def rows(arr):
    if mode == "outer":
      return np.repeat(arr, n_inputs, axis=0)          # channel = set * n_inputs + input
    return arr
  nbs = [s[0].shape[-1] for s in secs]
  nas = [s[1].shape[-1] for s in secs]
  bcat = np.concatenate([rows(s[0]) if per_channel else s[0] for s in secs], axis=-1)
  acat = np.concatenate([rows(s[1]) if per_channel else s[1] for s in secs], axis=-1)
  xo = x
  if mode == "outer":
    xo = np.tile(x, (1, n_sets)) if layout == "time" else np.tile(x, (n_sets, 1))
  ref = oracle.bank(nbs, nas, bcat, acat, xo, layout=layout, zero=zero)
  cut = int(rng.choice([0, N, 64, N // 2, int(rng.integers(0, N + 1))]))
  parts, used = [], set()
  for lo, hi in ((0, cut), (cut, N)):
    blk = x[lo:hi] if layout == "time" else x[:, lo:hi]
    if blk.size:
      parts.append(bank.process(np.ascontiguousarray(blk), layout=layout))
      used.update(bank.last_kernel.split("+"))
  got = np.concatenate(parts, axis=0 if layout == "time" else 1)
  for name in used:
    kernels[name] = kernels.get(name, 0) + 1
  # bit for bit, except that a NaN is a NaN (an unstable random recipe ends in inf - inf; x86 and
  # gfx950 then produce quiet NaNs of opposite sign bit, which Python cannot tell apart)
  same = (got.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(got) & np.isnan(ref))
  if not same.all():
    bad += 1
    print("MISMATCH case %d: kind=%s C=%d N=%d cut=%d layout=%s per_channel=%s mode=%s nb=%s na=%s kernels=%s maxdiff=%g"
          % (case, kind, C, N, cut, layout, per_channel, mode, nbs, nas, sorted(used), np.nanmax(np.abs(got - ref))))
print("%d cases, %d mismatches; kernels hit: %s" % (cases, bad, dict(sorted(kernels.items()))))
sys.exit(1 if bad else 0)
