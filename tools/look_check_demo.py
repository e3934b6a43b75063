"""The chunk-boundary check of the one-pass time-parallel kernel, shown at work (profiles/NOTES_r06.md 1).

Through a -DALZ_ABLATE build of alz_look.hip with ALZ_WAVE_DEBUG=4096 ONE chunk of the launch (channel group 1, the second
chunk of its third workgroup) is handed a start state that is off by 0.125 -- what a stale or torn published state
would do.  Expected: in the DEFERRED mode the next entry point raises and names CHUNK-BOUNDARY-CHECK-FAILED; in the
default mode the process call that launched the kernel notices, puts the bank's state back and processes the block again
in three launches: the caller gets a correct block (and look_stats counts one re-run).
    ALZ_LIBRARY=tools/variants/libalzhip_lookrace.so ALZ_WAVE_DEBUG=4096 python tools/look_check_demo.py"""
import os
import sys

sys.path.insert(0, '.')
import numpy as np
import torch
import audiolazy_amd as alz
from oracle import oracle
import bench

C, n = 512, 40 * 512
b, a = bench.resonator_coefs(4096)
pick = np.linspace(0, 4095, C).astype(int)
b, a = b[pick].copy(), a[pick].copy()
print("library %s, ALZ_WAVE_DEBUG=%s" % (os.environ.get("ALZ_LIBRARY", "(shipped)"), os.environ.get("ALZ_WAVE_DEBUG", "")))
for layout in ("time", "chan"):
  tm = layout == "time"
  ax = 0 if tm else 1
  rng = np.random.default_rng(11)
  x = rng.uniform(-1, 1, (n, C) if tm else (C, n))
  ref = oracle.bank([3], [3], b, a, x, layout=layout)
  den = np.abs(ref).max(axis=ax)
  for mode in ("deferred", "call"):
    bank = alz.FilterBank([(b, a)], n_inputs=C).set_time_parallel("one-pass").set_look_check(mode)
    bank.reset()
    y = bank.process(torch.from_numpy(x).cuda(), layout=layout)
    said = ""
    try:
      bank.sync()
    except RuntimeError as exc:
      said = str(exc)
    err = float((np.abs(y.cpu().numpy() - ref).max(axis=ax) / den).max())
    print("layout %-4s check %-8s: block error %.2e; %s; stats %s; kernels: %s"
          % (layout, mode, err, ("next entry point raised: ..." + said[said.find("waits that ran out"):][:120]) if said else "no error raised",
             bank.look_stats, bank.last_kernel[-110:]))
