#!/usr/bin/env python3
"""Differential fuzz of the ZFilter algebra against the imported reference (build container only):
``f(g)`` substitution, ``g / f``, ``k / f``, ``f ** -1``, ``f + g``, ``f * g``, ``f - g`` on seeded random
rational pairs.  Usage: python tools/fuzz_compose.py [n_cases] [seed]."""
import os
import random
import sys

sys.dont_write_bytecode = True
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, "/root/reference")

import audiolazy as ref            # noqa: E402
import audiolazy_amd as own        # noqa: E402
import gen_golden as gg            # noqa: E402


def outcome(mod, fn):
  try:
    r = fn()
  except Exception as exc:   # noqa: BLE001
    return ("raises", type(exc).__name__)
  if not isinstance(r, mod.ZFilter):
    return ("value", gg.typed(r))
  return ([(gg.typed(k), gg.typed(v)) for k, v in r.numpoly.terms()],
          [(gg.typed(k), gg.typed(v)) for k, v in r.denpoly.terms()])


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
  rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
  kinds = ["const", "delay", "zero", "fir", "rational", "rational", "rational", "fir"]
  exprs = {"f(g)": lambda f, g: f(g), "g/f": lambda f, g: g / f, "k/f": lambda f, g: 2.5 / f,
           "f**-1": lambda f, g: f ** -1, "f+g": lambda f, g: f + g, "f*g": lambda f, g: f * g,
           "f-g": lambda f, g: f - g, "3-f": lambda f, g: 3 - f, "f**2": lambda f, g: f ** 2}
  bad = {k: 0 for k in exprs}
  for _ in range(n):
    fn_, fd_ = gg.random_rational(rng, rng.choice(kinds))
    gn_, gd_ = gg.random_rational(rng, rng.choice(kinds))
    for name, e in exprs.items():
      a = outcome(ref, lambda: e(ref.ZFilter(dict(fn_), dict(fd_)), ref.ZFilter(dict(gn_), dict(gd_))))
      b = outcome(own, lambda: e(own.ZFilter(dict(fn_), dict(fd_)), own.ZFilter(dict(gn_), dict(gd_))))
      if a != b:
        bad[name] += 1
        if bad[name] <= 2:
          print(name, fn_, fd_, gn_, gd_, "\n  ref", a, "\n  own", b)
  print("cases", n, "differences", bad)
  return 1 if any(bad.values()) else 0


if __name__ == "__main__":
  sys.exit(main())
