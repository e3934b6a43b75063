#!/usr/bin/env python3
"""Differential fuzz of the ZFilter / CascadeFilter / ParallelFilter attribute surface against the imported reference on
FRESH random cases (build container only, no GPU): the probes of oracle/surface_probes.py (tests/golden/surface.json
pins 160 seeded cases of the same probes).  Usage: python tools/fuzz_surface.py [n_cases] [seed]."""
import os
import sys
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, "/root/reference")

import audiolazy as ref            # noqa: E402
import audiolazy_amd as own        # noqa: E402
import surface_probes as sp        # noqa: E402


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
  seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
  run = sp.filter_outcomes
  a, b = run(ref, seed, n), run(own, seed, n)
  bad = {}
  for i, (x, y) in enumerate(zip(a, b)):
    for name in x:
      if x[name] != y[name]:
        bad[name] = bad.get(name, 0) + 1
        if bad[name] <= 2:
          print(name, "case", i, "\n  ref", str(x[name])[:300], "\n  own", str(y[name])[:300])
  print("cases", n, "seed", seed, "differences", bad, "of probes", len(a[0]) if a else 0)
  return 1 if bad else 0


if __name__ == "__main__":
  sys.exit(main())
