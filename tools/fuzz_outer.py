#!/usr/bin/env python3
"""Differential fuzz of OUTER banks (filterbanks: every band on every input stream) against the oracle:
random band / stream counts -- including the few-stream banks that go through the input expansion
(csrc/alz_api.hip, k_expand) -- 1 to 4 cascaded sections with the curated tap patterns the streaming
and pipeline kernels take (and some they do not), both layouts, ragged block splits, an optional |x|
input stage, bit-exact or it prints the failing recipe.  Test infrastructure; run on the GPU box:
    python tools/fuzz_outer.py [cases] [seed]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import audiolazy_amd as alz
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4321)
# (numerator taps present, denominator taps present beyond a0): the patterns of the streaming kernels, and two outside them
PATTERNS = [((1, 0, 0), (1, 0)), ((1, 1, 0), (1, 0)), ((1, 0, 0), (1, 1)), ((1, 1, 0), (1, 1)), ((1, 0, 1), (1, 1)),
            ((1, 1, 1), (1, 1)), ((1, 0, 0), (0, 1)), ((1, 1, 1), (0, 0)), ((1, 1, 1, 1), (1, 1, 1)), ((0, 1, 0), (1, 1))]
bad = 0
kernels = {}
for case in range(cases):
  B = int(rng.choice([1, 2, 3, 4, 16, 22, 64, 100, 256]))
  S = int(rng.choice([1, 1, 2, 3, 5, 16, 17, 32, 64]))
  C = B * S
  if C > 8192:
    S = max(1, 8192 // B)
    C = B * S
  nsec = int(rng.choice([1, 1, 2, 4, 4]))
  same_pattern = bool(rng.integers(0, 2))
  N = int(rng.choice([1, 17, 64, 200, 1024, 2048 + 22, 4099]))
  layout = str(rng.choice(["time", "chan"]))
  use_abs = nsec == 1 and rng.random() < .25
  secs, nbs, nas, bs, as_ = [], [], [], [], []
  pat0 = PATTERNS[int(rng.integers(0, len(PATTERNS)))]
  for s in range(nsec):
    pb, pa = pat0 if same_pattern else PATTERNS[int(rng.integers(0, len(PATTERNS)))]
    nb, na = len(pb), len(pa) + 1
    b = rng.uniform(-1, 1, (B, nb)) * np.array(pb, dtype=float)
    a = np.concatenate([np.ones((B, 1)), rng.uniform(-.45, .45, (B, na - 1)) * np.array(pa, dtype=float)], axis=1)
    secs.append((b, a)); nbs.append(nb); nas.append(na)
    bs.append(np.repeat(b, S, axis=0)); as_.append(np.repeat(a, S, axis=0))
  zero = float(rng.choice([0., 0., .25]))
  bank = alz.FilterBank(secs, n_inputs=S, mode="outer")
  if use_abs:
    bank.set_input_map("abs")
  bank.reset(zero=zero)
  x = rng.uniform(-1, 1, (S, N))                       # [streams, samples]
  xin = np.abs(x) if use_abs else x
  ref = oracle.bank(nbs, nas, np.concatenate(bs, axis=1), np.concatenate(as_, axis=1), np.tile(xin, (B, 1)),
                    layout="chan", zero=zero)            # [B * S, N], channel = band * S + stream
  cut = int(rng.integers(0, N + 1))
  parts = []
  for lo, hi in ((0, cut), (cut, N)):
    blk = x[:, lo:hi]
    if blk.size:
      blk = np.ascontiguousarray(blk if layout == "chan" else blk.T)
      out = bank.process(blk, layout=layout)
      parts.append(out if layout == "chan" else out.T)
  got = np.ascontiguousarray(np.concatenate(parts, axis=1))
  for name in set(bank.last_kernel.split("+")):
    kernels[name] = kernels.get(name, 0) + 1
  same = (got.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(got) & np.isnan(ref))
  if not same.all():
    bad += 1
    print("MISMATCH case %d: B=%d S=%d nsec=%d N=%d cut=%d layout=%s nb=%s na=%s zero=%s abs=%s kernel=%s maxdiff=%g first=%s"
          % (case, B, S, nsec, N, cut, layout, nbs, nas, zero, use_abs, bank.last_kernel, np.nanmax(np.abs(got - ref)),
             np.argwhere(~same)[0].tolist()))
print("%d cases, %d mismatches; kernels: %s" % (cases, bad, kernels))
sys.exit(1 if bad else 0)
