"""The race behind round 5's one wrong block, made to happen on every launch (profiles/NOTES_r06.md 1).

Through a -DALZ_ABLATE build of alz_look.hip (tools/build_variant.sh lookrace alz_look.hip -DALZ_ABLATE), with
  ALZ_WAVE_DEBUG=1024   the workgroups of chunk 0 start ~40 us late (what a slow XCD gives once in a thousand launches)
  ALZ_WAVE_DEBUG=3072   the same, and the owner of the last chunk writes the bank's input history as soon as its last tile
                        has landed (round 5's order)
on the shape that failed (512 resonators, a 20480-sample block then a 3078-sample one = 6 chunks on 6 workgroups per
channel group), both layouts, in place and not.  Expected: 3072 -> the SECOND block wrong every time (chunk 0 reads the
history the launch itself has just written), silently; 1024 -> every block correct (the shipped order).
    ALZ_LIBRARY=tools/variants/libalzhip_lookrace.so ALZ_WAVE_DEBUG=3072 python tools/look_race_demo.py [reps]"""
import os
import sys

sys.path.insert(0, '.')
import numpy as np
import torch
import audiolazy_amd as alz
from oracle import oracle
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
b0, a0 = bench.resonator_coefs(4096)
print("library %s, ALZ_WAVE_DEBUG=%s" % (os.environ.get("ALZ_LIBRARY", "(shipped)"), os.environ.get("ALZ_WAVE_DEBUG", "")))
# 512 channels, 6 chunks: 6 workgroups per channel group, workgroup ids 6 g (chunk 0) and 6 g + 5 (the last chunk) -- never on the
# same XCD (id mod 8), so the early write stays in the writer's L2 and chunk 0 reads the old words from memory: no error
# even with the round-5 order.  256 channels, 9 chunks: 9 workgroups per group, ids 9 g and 9 g + 8 -- the SAME XCD, one L2.
for C, n2, layout in ((512, 6 * 512 + 6, "chan"), (512, 6 * 512 + 6, "time"), (256, 9 * 512 + 6, "chan"), (256, 9 * 512 + 6, "time")):
  pick = np.linspace(0, 4095, C).astype(int)
  b, a = b0[pick].copy(), a0[pick].copy()
  tm = layout == "time"
  ax = 0 if tm else 1
  n1 = 40 * 512
  rng = np.random.default_rng(n1 + n2)
  x1 = rng.uniform(-1, 1, (n1, C) if tm else (C, n1))
  x2 = rng.uniform(-1, 1, (n2, C) if tm else (C, n2))
  ref = oracle.bank([3], [3], b, a, np.concatenate([x1, x2], axis=ax), layout=layout)
  refs = (ref[:n1], ref[n1:]) if tm else (ref[:, :n1], ref[:, n1:])
  for inplace in (True, False):
    bad, silent, worst = [0, 0], [0, 0], [0.0, 0.0]
    for rep in range(reps):
      bank = alz.FilterBank([(b, a)], n_inputs=C).set_time_parallel("one-pass").set_look_check("deferred")
      bank.reset()
      for k, x in enumerate((x1, x2)):
        xd = torch.from_numpy(x).cuda()
        y = bank.process(xd, layout=layout, out=xd if inplace else None)
        reported = False
        try:
          bank.sync()
        except RuntimeError:
          reported = True
        yh = y.cpu().numpy()
        den = np.abs(refs[k]).max(axis=ax)
        e = float((np.abs(yh - refs[k]).max(axis=ax) / den).max())
        worst[k] = max(worst[k], e)
        if not e <= 1e-8:
          bad[k] += 1
          silent[k] += not reported
        if reported:
          break
    print("%d channels, second block %d samples, layout %-4s inplace %-5s: %d reps; wrong blocks: first %d (silent %d), second %d (silent %d); worst error %.2e / %.2e"
          % (C, n2, layout, inplace, reps, bad[0], silent[0], bad[1], silent[1], worst[0], worst[1]))
