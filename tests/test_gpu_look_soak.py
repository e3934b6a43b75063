"""Soak of the one-pass time-parallel kernel's cross-workgroup protocol (csrc/alz_look.hip), in the suite.

Round 5 saw ONE wrong block in about a thousand runs of the one-pass cases -- 512 resonators, channel-major, in place,
a 20480-sample block followed by a 3078-sample one; the second block came back with a normalised error of 0.76 -- and
could not say whether the kernel had reported a timed-out wait (nobody looked) or had been silently wrong.  Round 6 found
the cause by reading the protocol (profiles/NOTES_r06.md 1): the owner of a block's LAST chunk wrote the bank's input
history for the next block as soon as its last tile had landed, while the workgroup of chunk 0 reads the same words when
it STARTS -- and with one chunk per workgroup (the 3078-sample block: 6 chunks on 6 workgroups per channel group) nothing
orders the two.  The kernel now writes that history after its last store, which is ordered behind chunk 0's read through
the published chunk states (and the output state behind a flag of the CHAIN wave).

This file is the evidence asked for: the failing shape and its time-major twin, 3000 iterations each (6000 launches per
layout, a quarter of them in place on one-chunk-per-workgroup blocks), every second iteration beside foreign work
(tests/helpers/hog.hip holding CUs for 0.05 - 2 ms on another stream), each block classified as
  correct / gave-up (the kernel reported a wait that ran out: WHICH one) / wrong-and-silent
in the DEFERRED check mode, so that the kernel's own report is what is counted (the default mode would hide a give-up by
processing the block again).  Asserted: zero of the last two.  The counts go to gpurun_out/look_soak.log when that
directory exists (copied to profiles/r06_look_soak.log)."""
import ctypes
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITER = int(os.environ.get("ALZ_LOOK_SOAK_ITER", "3000"))


def _hog():
  path = os.path.join(ROOT, "tests", "helpers", "libhog.so")
  assert os.path.exists(path), "tests/helpers/libhog.so is built by __graft_entry__.build()"
  lib = ctypes.CDLL(path)
  lib.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
  lib.hog_launch.restype = ctypes.c_int
  return lib


def _log(line):
  out = os.path.join(ROOT, "gpurun_out")
  print(line)
  if os.path.isdir(out):
    with open(os.path.join(out, "look_soak.log"), "a") as f:
      f.write(line + "\n")


@pytest.mark.parametrize("layout", ["chan", "time"])
def test_one_pass_soak(layout):
  import torch
  import audiolazy_amd as alz
  import bench
  from oracle import oracle
  C, n1, n2 = 512, 40 * 512, 6 * 512 + 6
  tm = layout == "time"
  ax = 0 if tm else 1
  b, a = bench.resonator_coefs(4096)
  pick = np.linspace(0, 4095, C).astype(int)
  b, a = b[pick].copy(), a[pick].copy()
  rng = np.random.default_rng(20480 + 3078)
  x1 = rng.uniform(-1, 1, (n1, C) if tm else (C, n1))
  x2 = rng.uniform(-1, 1, (n2, C) if tm else (C, n2))
  ref = oracle.bank([3], [3], b, a, np.concatenate([x1, x2], axis=ax), layout=layout)
  refs = [torch.from_numpy(np.ascontiguousarray(r)).cuda() for r in ((ref[:n1], ref[n1:]) if tm else (ref[:, :n1], ref[:, n1:]))]
  scale = [r.abs().amax(dim=ax).clamp_min(1e-300) for r in refs]
  xs = [torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()]
  hog = _hog()
  side = torch.cuda.Stream()
  hrng = np.random.default_rng(7)
  bank = alz.FilterBank([(b, a)], n_inputs=C).set_time_parallel("one-pass").set_look_check("deferred")
  counts = {}
  sites = {}
  examples = []
  t0 = time.time()
  for it in range(ITER):
    inplace = (it >> 1) & 1
    foreign = it & 1
    bank.reset()
    for k in range(2):
      work = xs[k].clone() if inplace else xs[k]
      if foreign:
        # 16 - 128 CUs held for 0.05 - 2 ms with most of their LDS: the one-pass workgroups of those CUs start late
        assert hog.hog_launch(int(hrng.integers(16, 129)), 110 * 1024, int(hrng.integers(50, 2001)), side.cuda_stream) == 0
      y = bank.process(work, layout=layout, out=work if inplace else None)
      assert "k_look" in bank.last_kernel, bank.last_kernel
      gave_up = None
      try:
        bank.sync()
      except RuntimeError as exc:
        gave_up = str(exc)
      err = float(((y - refs[k]).abs().amax(dim=ax) / scale[k]).max())
      good = err <= 1e-8
      what = "gave-up" if gave_up else ("correct" if good else "WRONG-AND-SILENT")
      key = (what, "block%d" % (k + 1), "inplace" if inplace else "out-of-place", "foreign" if foreign else "idle")
      counts[key] = counts.get(key, 0) + 1
      if gave_up:
        for s in bank.look_stats["last_sites"]:
          sites[s] = sites.get(s, 0) + 1
      if what != "correct" and len(examples) < 8:
        examples.append((it, key, err, gave_up))
      if gave_up:
        break                                     # (the bank's state is invalid: next iteration starts from reset)
    if foreign and (it & 63) == 1:
      torch.cuda.synchronize()                    # (do not let the side stream's queue grow without bound)
  torch.cuda.synchronize()
  _log("one-pass soak, layout %s: %d iterations x 2 blocks (512 ch; %d then %d samples) in %.1f s" % (layout, ITER, n1, n2, time.time() - t0))
  for key in sorted(counts):
    _log("  %-17s %s %-12s %-7s %6d" % (key + (counts[key],)))
  if sites:
    _log("  waits that ran out: %s" % sites)
  for e in examples:
    _log("  example: iteration %d %s error %.3g %s" % e)
  bad = {k: v for k, v in counts.items() if k[0] != "correct"}
  assert not bad, (bad, sites, examples)
