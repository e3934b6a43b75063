"""Time-varying filters on the GPU (alz_tv_process_dev) against reference-generated vectors
(tests/golden/timevar.json) and the oracle: bit-exact."""
import numpy as np
import pytest

from conftest import load_golden, unhex
from oracle import oracle
from test_timevar_cpu import coefs, normalised, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def al():
  import audiolazy_amd
  assert audiolazy_amd.device_count() >= 1
  return audiolazy_amd


@pytest.mark.parametrize("idx", range(8))
def test_stream_coefficients_match_reference(al, idx):
  g = load_golden("timevar.json")
  c = g["direct"][idx]
  mk = lambda lst: [al.Stream(v) if isinstance(v, list) else v for v in lst]
  kw = dict(zero=float.fromhex(c["zero"]))
  if c["memory"] is not None:
    kw["memory"] = unhex(c["memory"])
  filt = al.ZFilter(mk(coefs(c["b"])), mk(coefs(c["a"])))
  assert same_bits(list(filt(unhex(g["x"]), **kw)), unhex(c["y"]))


@pytest.mark.parametrize("idx", range(10))
def test_designs_called_with_streams(al, idx):
  g = load_golden("timevar.json")
  c = g["designs"][idx]
  args = [al.Stream(unhex(v)) if isinstance(v, list) else float.fromhex(v["const"]) for v in c["args"]]
  family, strategy = c["name"].split(".")
  filt = getattr(getattr(al, family), strategy)(*args)
  assert same_bits(list(filt(unhex(g["x"]))), unhex(c["y"])), c["name"]


def test_blocks_continue_one_stream(al):
  """Coefficient streams and state carry over block boundaries (block=37 vs one block)."""
  from audiolazy_amd import timevar
  rng = np.random.default_rng(2)
  N = 1000
  x, b0, a1, a2 = rng.uniform(-1, 1, N), rng.uniform(.1, 1, N), rng.uniform(-.9, .9, N), rng.uniform(-.4, .4, N)
  ref = oracle.tv_df1([b0, .5], [1., a1, a2], x, memory=[.2, -.1], zero=.3)
  got = list(timevar.run([iter(b0), .5], [1., iter(a1), iter(a2)], iter(x), memory=[.2, -.1], zero=.3, block=37))
  assert same_bits(got, ref)
  # a coefficient stream shorter than the input ends the output there
  got = list(timevar.run([iter(b0[:123])], [1., iter(a1)], iter(x), block=50))
  assert same_bits(got, oracle.tv_df1([b0[:123]], [1., a1[:123]], x[:123]))


def test_errors(al):
  from audiolazy_amd import timevar
  with pytest.raises(ZeroDivisionError):                    # lazy_filters.py:177-178
    list(timevar.run([iter([1., 2.])], [0., .5], [1., 2.]))
  with pytest.raises(ValueError):                           # ... which ZFilter turns into z / .5 first (:126-132)
    list(al.ZFilter([al.Stream([1., 2.])], [0., .5])([1., 2.]))
  with pytest.raises(ValueError):                           # lazy_filters.py:165-168
    (al.ZFilter([al.Stream([1., 2.])]) * al.z)([1., 2.])
  with pytest.raises(NotImplementedError):                  # beyond the 17-tap register window
    list(al.ZFilter([al.Stream([1., 2.])] + [0.] * 16 + [1.])([1., 2.]))


@pytest.mark.parametrize("layout", ["time", "chan"])
def test_multichannel_block(al, layout):
  """Many channels in one launch: a series shared by all channels and a per-channel one."""
  import torch
  from audiolazy_amd import timevar
  rng = np.random.default_rng(7)
  N, C = 300, 70
  x = rng.uniform(-1, 1, (N, C))
  b0 = rng.uniform(.1, 1, N)                # shared
  a1 = rng.uniform(-.9, .9, (N, C))         # per channel
  a2 = rng.uniform(-.3, .3, N)
  xt, a1t = (x, a1) if layout == "time" else (x.T.copy(), a1.T.copy())
  dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
  xh = torch.full((1, C), .25, dtype=torch.float64, device="cuda")
  yh = torch.full((2, C), .25, dtype=torch.float64, device="cuda")
  half = N // 2

  def part(lo, hi):
    sl = (slice(lo, hi), slice(None)) if layout == "time" else (slice(None), slice(lo, hi))
    return timevar.process_block([dev(b0[lo:hi]), -.5], [2., dev(a1t[sl]), dev(a2[lo:hi])], dev(xt[sl]),
                                 xh=xh, yh=yh, zero=.25, layout=layout).cpu().numpy()
  y = np.concatenate([part(0, half), part(half, N)], axis=0 if layout == "time" else 1)
  if layout == "chan":
    y = y.T
  for ch in (0, 1, 33, 69):
    ref = oracle.tv_df1([b0, -.5], [2., a1[:, ch], a2], x[:, ch], zero=.25)
    assert same_bits(y[:, ch], ref), ch
