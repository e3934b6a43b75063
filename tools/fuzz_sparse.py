#!/usr/bin/env python3
"""Differential fuzz of round 6's kernels against the oracle, bit for bit: the comb family (k_comb_tm / k_comb_cm / k_string /
k_sparse: feedback and feed-forward combs, linearize()d pairs, numerator pairs and triples in front of them) and the middle
shapes (k_mid: dense IIR sections of order 3 - 8, all-pole, long numerators in front of one or two poles, a far tap in front of
a pole -- maverage.recursive), on random channel counts (whole and ragged groups, single strings), both layouts, in place
where the engine allows it, shared / per-channel coefficients, random delay-line contents, streams cut into random blocks
(shorter than a step, a tile, a chunk, the delay line).  Prints the failing recipe; run on a GPU box:
    python tools/fuzz_sparse.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import audiolazy_amd as alz
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
bad = 0
kernels = {}


def same_bits(a, b):
  return a.shape == b.shape and bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


def stable_den(C, K):
  out = np.zeros((C, K + 1))
  for c in range(C):
    poles = []
    for _ in range(K // 2):
      r, w = rng.uniform(0.4, 0.97), rng.uniform(0.05, 3.0)
      poles += [r * np.exp(1j * w), r * np.exp(-1j * w)]
    if K % 2:
      poles.append(rng.uniform(-0.9, 0.95))
    out[c] = np.real(np.poly(poles))
    out[c, 0] = 1.0
  return out


for case in range(cases):
  fam = str(rng.choice(["fb", "lin", "ff", "fflin", "mixed", "dense", "allpole", "longnum", "far"]))
  C = int(rng.choice([1, 1, 2, 16, 17, 32, 48, 64, 80, 272, 300]))
  layout = str(rng.choice(["time", "chan"]))
  per_channel = bool(rng.integers(0, 2)) or fam in ("dense", "allpole", "longnum")
  g = rng.uniform(0.2, 0.97, C) * rng.choice([-1.0, 1.0], C)
  f = rng.uniform(0.1, 0.9, C)
  D = int(rng.choice([16, 17, 31, 63, 64, 65, 100, 128, 255, 441, 512, 513, 700, 1500]))
  if fam == "fb":
    b = np.ones((C, 1)); a = np.zeros((C, D + 1)); a[:, 0] = 1; a[:, D] = -g
  elif fam == "lin":
    b = np.ones((C, 1)); a = np.zeros((C, D + 2)); a[:, 0] = 1; a[:, D] = -g * (1 - f); a[:, D + 1] = -g * f
  elif fam == "ff":
    b = np.zeros((C, D + 1)); b[:, 0] = 1; b[:, D] = g; a = np.ones((C, 1))
  elif fam == "fflin":
    b = np.zeros((C, D + 2)); b[:, 0] = 1; b[:, D] = g * (1 - f); b[:, D + 1] = g * f; a = np.ones((C, 1))
  elif fam == "mixed":
    k2 = int(rng.integers(1, 9))
    b = np.zeros((C, k2 + 1)); b[:, 0] = rng.uniform(0.5, 1.5, C); b[:, k2] = rng.uniform(-0.5, 0.5, C)
    a = np.zeros((C, D + 2)); a[:, 0] = 1; a[:, D] = -g * 0.6
    if rng.random() < 0.5:
      a[:, D + 1] = -g * 0.3
    else:
      a = a[:, :D + 1]
  elif fam == "dense":
    K = int(rng.integers(3, 9))
    b = rng.uniform(-1, 1, (C, K + 1)); a = stable_den(C, K)
  elif fam == "allpole":
    K = int(rng.integers(3, 9))
    b = rng.uniform(0.1, 1, (C, 1)); a = stable_den(C, K)
  elif fam == "longnum":
    K = int(rng.integers(1, 3))
    b = rng.uniform(-1, 1, (C, int(rng.integers(4, 10)))); a = stable_den(C, K)
  else:
    S = int(rng.choice([9, 10, 63, 64, 65, 200, 256]))
    nbd = int(rng.integers(1, 3))
    b = np.zeros((C, S + 1)); b[:, :nbd] = rng.uniform(0.2, 1, (C, nbd)); b[:, S] = -rng.uniform(0.2, 1, C)
    K = int(rng.integers(1, 3))
    a = stable_den(C, K)
    a[:, 1:] *= 0.98
  if not per_channel:
    b, a = b[0].copy(), a[0].copy()
  nb, na = b.shape[-1], a.shape[-1]
  n_blocks = int(rng.integers(1, 4))
  lens = [int(rng.choice([2, 6, 20, 64, 66, 130, 256, 258, 600, 1024, 1500, 3 * D + 2])) for _ in range(n_blocks)]
  if layout == "chan" and C > 1:
    lens = [m + (m & 1) for m in lens]           # (even rows keep the 16-byte kernels eligible; odd ones are covered by the suite)
  tm = layout == "time"
  ax = 0 if tm else 1
  xs = [rng.uniform(-1, 1, (m, C) if tm else (C, m)) for m in lens]
  xh0 = rng.uniform(-1, 1, (C, max(nb - 1, 1)))
  yh0 = rng.uniform(-1, 1, (C, max(na - 1, 1)))
  inplace = nb == 1 and rng.random() < 0.4
  try:
    bank = alz.FilterBank([(b, a)], n_inputs=C)
    bank.set_state(xh0, yh0)
    ref = oracle.bank([nb], [na], b, a, np.concatenate(xs, axis=ax), layout=layout, xh=xh0.copy(), yh=yh0.copy())
    at, ok = 0, True
    for x in xs:
      xd = torch.from_numpy(x).cuda()
      y = bank.process(xd, layout=layout, out=xd if inplace else None).cpu().numpy()
      for k in bank.last_kernel.split("+"):
        kernels[k] = kernels.get(k, 0) + 1
      m = x.shape[ax]
      r = ref[at:at + m] if tm else ref[:, at:at + m]
      ok = ok and same_bits(y, r)
      at += m
    if ok:
      xh1, yh1 = bank.get_state()
      # the state the bank is left with continues the stream: one more block
      x = rng.uniform(-1, 1, (40, C) if tm else (C, 40))
      whole = oracle.bank([nb], [na], b, a, np.concatenate(xs + [x], axis=ax), layout=layout, xh=xh0.copy(), yh=yh0.copy())
      y = bank.process(torch.from_numpy(x).cuda(), layout=layout).cpu().numpy()
      ok = same_bits(y, whole[at:] if tm else whole[:, at:])
  except Exception as exc:                          # an engine error is a failure too
    ok = False
    print("case %d raised %r" % (case, exc))
  if not ok:
    bad += 1
    print("MISMATCH case %d: family %s C %d layout %s per_channel %s D %d nb %d na %d blocks %s inplace %s kernels %s"
          % (case, fam, C, layout, per_channel, D, nb, na, lens, inplace, bank.last_kernel))
print("fuzz_sparse: %d cases, %d mismatches; kernels: %s" % (cases, bad, dict(sorted(kernels.items()))))
sys.exit(1 if bad else 0)
