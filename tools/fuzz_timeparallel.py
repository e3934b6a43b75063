#!/usr/bin/env python3
"""Differential fuzz of the opt-in time-parallel mode against the oracle (round 5: dot-product zero-state pass, DPP chunk-state
recursion, time-major cascades, one-pass form in both layouts / in place / through |x|).  Random shapes chosen so that every
form gets its share -- fused cascades of 2 - 4 two-pole sections on OUTER banks of 1 / 2 / 64 / 128 streams in both layouts
(k_cdot or the cascade kernel for the zero-state pass, k_pipe or k_casc for the replay), single sections on diagonal banks
(k_look, the three-launch form) --, two or three consecutive blocks, optional arbitrary or self-consistent set_state, optional
in place.  Tolerance 1e-8 normalised per channel (the mode's bar; 1e-6 is the contract); prints the failing recipe.
    python tools/fuzz_timeparallel.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import audiolazy_amd as alz
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 505)
NUM = [(1, 0, 0), (1, 1, 0), (1, 0, 1), (1, 1, 1)]            # numerator taps present (at most three: the fused form's shapes)
bad = 0
kernels = {}


def norm_err(got, ref):
  den = np.abs(ref).max(axis=1)
  den[den == 0] = 1.0
  return float((np.abs(got - ref).max(axis=1) / den).max())


for case in range(cases):
  cascade = rng.random() < .6
  layout = str(rng.choice(["time", "chan"]))
  if cascade:
    B = int(rng.choice([4, 8, 16, 64, 128, 256]))
    S = int(rng.choice([1, 1, 1, 2, 64, 128]))
    while B * S > 4096:                                  # (the oracle is one CPU thread: keep a case under a second or two)
      B //= 2
    nsec = int(rng.choice([2, 3, 4, 4]))
    pats = [NUM[int(rng.integers(0, 4))] for _ in range(nsec)]
    if rng.random() < .5:
      pats = [pats[0]] * nsec
    mode = "outer"
  else:
    B = int(rng.choice([64, 256, 512, 1024, 2048]))
    S, nsec, mode = B, 1, "diagonal"
    pats = [NUM[int(rng.integers(0, 4))]]
  C = B * S if cascade else B
  secs, nbs, nas, bs, as_ = [], [], [], [], []
  for pb in pats:
    r, th = rng.uniform(.6, .999, B), rng.uniform(.02, 3., B)
    nb = max(i + 1 for i, v in enumerate(pb) if v)
    b = (rng.uniform(-1, 1, (B, 3)) * np.array(pb, dtype=float))[:, :nb]
    a = np.stack([np.ones(B), -2 * r * np.cos(th), r * r], axis=1)
    secs.append((b, a)); nbs.append(nb); nas.append(3)
    rep = S if cascade else 1
    bs.append(np.repeat(b, rep, axis=0)); as_.append(np.repeat(a, rep, axis=0))
  tp = rng.choice(["auto", "auto", "one-pass", 1024, 4096]) if not cascade else rng.choice(["auto", "auto", 256, 1024])
  tp = True if tp == "auto" else (tp if tp == "one-pass" else int(tp))
  n_in = S if cascade else C
  bank = alz.FilterBank(secs, n_inputs=n_in, mode=mode).set_time_parallel(tp)
  use_abs = (not cascade) and rng.random() < .3
  inplace = (not cascade) and rng.random() < .3
  if use_abs:
    bank.set_input_map("abs")
  bank.reset()
  thx, thy = sum(nb - 1 for nb in nbs), 2 * nsec
  xh, yh = np.zeros((C, max(thx, 1))), np.zeros((C, thy))
  state = str(rng.choice(["reset", "reset", "arbitrary", "consistent"]))
  if state != "reset":
    xh, yh = rng.uniform(-1e-2, 1e-2, xh.shape), rng.uniform(-1e-2, 1e-2, yh.shape)
    if use_abs:
      xh = np.abs(xh)                                   # (the bank keeps its input history mapped)
    if state == "consistent":
      ox, oy = nbs[0] - 1, 0
      for s in range(1, nsec):
        xh[:, ox:ox + nbs[s] - 1] = yh[:, oy:oy + nbs[s] - 1]
        ox += nbs[s] - 1
        oy += 2
    bank.set_state(xh[:, :thx] if thx else xh, yh)
  lens = [int(rng.choice([1 << 13, 1 << 14, 3 << 12, 40 * 512, 1 << 15, (1 << 14) + 64])) for _ in range(int(rng.integers(2, 4)))]
  if C * nsec >= 4096:
    lens = [min(m, 1 << 14) for m in lens[:2]]
  xs = [rng.uniform(-1, 1, (n_in, m)) for m in lens]
  xall = np.concatenate(xs, axis=1)
  xin = np.abs(xall) if use_abs else xall
  ref = oracle.bank(nbs, nas, np.concatenate(bs, axis=1), np.concatenate(as_, axis=1),
                    np.tile(xin, (B, 1)) if cascade else xin, layout="chan", xh=xh.copy(), yh=yh.copy())
  at, worst, names = 0, 0.0, []
  try:
    for x in xs:
      blk = torch.from_numpy(np.ascontiguousarray(x if layout == "chan" else x.T)).cuda()
      out = bank.process(blk, layout=layout, out=blk if inplace else None).cpu().numpy()
      out = out if layout == "chan" else out.T
      m = x.shape[1]
      worst = max(worst, norm_err(out, ref[:, at:at + m]))
      names.append(bank.last_kernel)
      at += m
  except Exception as exc:
    worst = float("inf")
    names.append("EXCEPTION %s" % exc)
  for nm in names:
    for part in set(nm.split("+")):
      kernels[part] = kernels.get(part, 0) + 1
  if not worst <= 1e-8:
    bad += 1
    print("MISMATCH case %d: cascade=%s B=%d S=%d nsec=%d pats=%s layout=%s tp=%s abs=%s inplace=%s state=%s lens=%s err=%.3g kernels=%s"
          % (case, cascade, B, S, nsec, pats, layout, tp, use_abs, inplace, state, lens, worst, names))
print("%d cases, %d mismatches; kernels: %s" % (cases, bad, kernels))
sys.exit(1 if bad else 0)
