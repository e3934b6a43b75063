"""Stress of the one-pass time-parallel form on the shape that failed once in the gpu suite (512 resonators, a 20480-sample
block then a 3078-sample one): every combination of layout x in place, many repetitions, failures counted per block.
    python tools/look_stress.py [reps]"""
import sys

sys.path.insert(0, '.')
import numpy as np
import torch
import audiolazy_amd as alz
from oracle import oracle
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
C = 512
b, a = bench.resonator_coefs(4096)
pick = np.linspace(0, 4095, C).astype(int)
b, a = b[pick].copy(), a[pick].copy()


def norm_err(got, ref, axis):
  den = np.abs(ref).max(axis=axis)
  den[den == 0] = 1.0
  return float((np.abs(got - ref).max(axis=axis) / den).max())


for layout in ("chan", "time"):
  tm = layout == "time"
  ax = 0 if tm else 1
  for n1, n2 in ((40 * 512, 6 * 512 + 6), (6 * 512 + 6, 40 * 512), (8 * 512, 8 * 512)):
    rng = np.random.default_rng(n1 + n2)
    x1 = rng.uniform(-1, 1, (n1, C) if tm else (C, n1))
    x2 = rng.uniform(-1, 1, (n2, C) if tm else (C, n2))
    ref = oracle.bank([3], [3], b, a, np.concatenate([x1, x2], axis=ax), layout=layout)
    r1, r2 = (ref[:n1], ref[n1:]) if tm else (ref[:, :n1], ref[:, n1:])
    for inplace in (True, False):
      bad = [0, 0]
      worst = [0.0, 0.0]
      chans = set()
      for rep in range(reps):
        bank = alz.FilterBank([(b, a)], n_inputs=C).set_time_parallel("one-pass")
        bank.reset()
        for k, (x, r) in enumerate(((x1, r1), (x2, r2))):
          xd = torch.from_numpy(x).cuda()
          y = bank.process(xd, layout=layout, out=xd if inplace else None).cpu().numpy()
          e = norm_err(y, r, ax)
          worst[k] = max(worst[k], e)
          if not e <= 1e-8:
            bad[k] += 1
            den = np.abs(r).max(axis=ax)
            per = np.abs(y - r).max(axis=ax) / den
            chans.update(np.nonzero(per > 1e-8)[0].tolist()[:8])
        del bank
      print("layout %-4s blocks %5d + %5d inplace %-5s: %d reps, failures block1 %d block2 %d, worst %.2e / %.2e, channels %s"
            % (layout, n1, n2, inplace, reps, bad[0], bad[1], worst[0], worst[1], sorted(chans)[:12]))
