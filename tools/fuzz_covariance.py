"""Differential fuzz of the covariance-method callers against the imported reference (build container only):
acorr, lag_matrix, lpc.covar, lpc.kcovar, parcor, parcor_stable on seeded random blocks (floats, ints, mixed, degenerate).
No GPU here: float blocks take their lag matrix from the C oracle (the engine's is held to it bit for bit by
tests/test_gpu_covariance.py), integer blocks run the host arithmetic as shipped.
Usage: python tools/fuzz_covariance.py [n_cases] [seed]."""
import importlib, random, sys
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference"); sys.path.insert(0, ".")
import numpy as np
import audiolazy as ref
import audiolazy_amd as own
from oracle import oracle

mod = importlib.import_module("audiolazy_amd.lpc")
mod.lag_matrix_frames = lambda sig, frame_len, max_lag, hop=None, device=0: oracle.lag_matrix(np.asarray(sig, dtype=np.float64), max_lag)[None]
mod.acorr_frames = lambda sig, frame_len, max_lag, hop=None, device=0: oracle.acorr(np.asarray(sig, dtype=np.float64), max_lag)[None]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def block():
  n = rng.choice([3, 4, 5, 8, 13, 30, 64])
  kind = rng.choice(["float", "float", "int", "mixed", "periodic", "sparse"])
  if kind == "float": return [rng.uniform(-1, 1) for _ in range(n)]
  if kind == "int": return [rng.randint(-9, 9) for _ in range(n)]
  if kind == "mixed": return [rng.choice([rng.randint(-4, 4), rng.uniform(-2, 2)]) for _ in range(n)]
  if kind == "periodic": return ([-1., 0., 1., 0.] * n)[:n]
  return [rng.choice([0., 0., rng.uniform(-1, 1)]) for _ in range(n)]


def outcome(fn):
  try:
    r = fn()
  except Exception as exc:  # noqa: BLE001
    return ("raises", type(exc).__name__, str(exc))
  if hasattr(r, "numlist"):
    return ("filter", [(type(v).__name__, float(v).hex()) for v in r.numlist], [float(v).hex() for v in r.denlist],
            type(r.error).__name__, float(r.error).hex())
  return ("value", repr(r))


bad = {}
for case in range(N):
  blk = block()
  order = rng.choice([None, 0, 1, 2, 3, 5, len(blk) - 1, len(blk)])
  probes = {
    "lag_matrix": lambda m: m.lag_matrix(list(blk), order),
    "acorr": lambda m: m.acorr(list(blk), order if order is None else order + case % 3),
    "covar": lambda m: m.lpc.covar(list(blk), order if order else 2),
    "kcovar": lambda m: m.lpc.kcovar(list(blk), order if order else 2),
    "parcor": lambda m: [float(k).hex() for k in m.parcor(m.ZFilter([1.] + [rng2.uniform(-.9, .9) for _ in range(nt)]))],
    "parcor_stable": lambda m: m.parcor_stable(1 / m.ZFilter([1.] + [rng2.uniform(-1.2, 1.2) for _ in range(nt)])),
    "toeplitz": lambda m: m.toeplitz(list(blk)),
  }
  for name, p in probes.items():
    nt = 1 + case % 6
    rng2 = random.Random(case); a = outcome(lambda: p(ref))
    rng2 = random.Random(case); b = outcome(lambda: p(own))
    if a != b:
      bad[name] = bad.get(name, 0) + 1
      if bad[name] <= 3: print(name, blk, order, "\n  ref", str(a)[:300], "\n  own", str(b)[:300])
print("cases", N, "differences", bad)
