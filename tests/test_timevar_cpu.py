"""Time-varying filters on the CPU side: the oracle restatement (oracle.tv_df1, a pure-Python
loop following lazy_filters.py:197-257) against reference-generated vectors
(tests/golden/timevar.json, made by oracle/gen_golden.py with Stream coefficients and with the
designs called on Streams), and the host logic that builds the coefficient streams."""
import numpy as np
import pytest

from conftest import load_golden, unhex
from oracle import oracle


def coefs(spec):
  return [unhex(c["series"]) if "series" in c else float.fromhex(c["const"]) for c in spec]


def normalised(b, a):
  """The reference's rewrite of a series a0 (lazy_filters.py:166-174): everything times 1 / a0."""
  if not isinstance(a[0], list):
    return b, a
  inv = [1 / v for v in a[0]]

  def mul(v):
    if isinstance(v, list):
      return [p * i for p, i in zip(v, inv)]
    return [v * i for i in inv] if v != 0 else 0.
  return [mul(v) for v in b], [1.] + [mul(v) for v in a[1:]]


def same_bits(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return a.shape == b.shape and bool(np.all(a.view(np.uint64) == b.view(np.uint64)))


@pytest.mark.parametrize("idx", range(8))
def test_oracle_direct_stream_coefficients(idx):
  g = load_golden("timevar.json")
  c = g["direct"][idx]
  b, a = normalised(coefs(c["b"]), coefs(c["a"]))
  memory = None if c["memory"] is None else unhex(c["memory"])
  y = oracle.tv_df1(b, a, unhex(g["x"]), memory=memory, zero=float.fromhex(c["zero"]))
  assert same_bits(y, unhex(c["y"]))


@pytest.mark.parametrize("idx", range(10))
def test_designs_on_streams_give_the_reference_coefficients(idx):
  import audiolazy_amd as al
  from audiolazy_amd.stream import Stream
  g = load_golden("timevar.json")
  c = g["designs"][idx]
  b, a = coefs(c["b"]), coefs(c["a"])
  assert same_bits(oracle.tv_df1(b, a, unhex(g["x"])), unhex(c["y"]))
  args = [unhex(v) if isinstance(v, list) else float.fromhex(v["const"]) for v in c["args"]]
  family, strategy = c["name"].split(".")
  filt = getattr(getattr(al, family), strategy)(*[Stream(v) if isinstance(v, list) else v for v in args])
  assert not filt.is_lti() and filt.is_causal()
  mine = filt.numlist + filt.denlist
  assert (len(filt.numlist), len(filt.denlist)) == (len(b), len(a))
  for got, ref in zip(mine, b + a):
    if isinstance(ref, list):
      assert same_bits(list(got), ref), c["name"]
    elif hasattr(got, "__iter__"):    # a coefficient the reference keeps constant: a constant series here
      vals = list(got)
      assert len(vals) == len(unhex(g["x"])) and all(v == ref for v in vals), c["name"]
    else:
      assert got == ref, c["name"]


def test_lti_filters_stay_lti():
  import audiolazy_amd as al
  s, Hz = al.sHz(48000)
  assert al.lowpass.pole(1000 * Hz).is_lti()
  assert al.resonator.z_exp(1000 * Hz, 100 * Hz).is_lti()
  assert al.CascadeFilter(al.lowpass.pole(.1), al.highpass.z(.2)).is_lti()
  tv = al.ZFilter([al.Stream([.1, .2])], [1., .5])
  assert not tv.is_lti()
  assert not al.CascadeFilter(al.lowpass.pole(.1), tv).is_lti()


def as_arrays(coefs_):
  return [np.asarray(v, dtype=np.float64) if isinstance(v, list) else v for v in coefs_]


@pytest.mark.parametrize("idx", range(8))
def test_c_oracle_direct_stream_coefficients(idx):
  """alzo_tv_df1 (oracle/alz_oracle.c, what the full-extent GPU comparisons use) pinned on the same
  reference-generated vectors as the pure-Python form."""
  g = load_golden("timevar.json")
  c = g["direct"][idx]
  b, a = normalised(coefs(c["b"]), coefs(c["a"]))
  memory = None if c["memory"] is None else unhex(c["memory"])
  y = oracle.tv_df1_c(as_arrays(b), as_arrays(a), unhex(g["x"]), memory=memory, zero=float.fromhex(c["zero"]))
  assert same_bits(y, unhex(c["y"]))


@pytest.mark.parametrize("idx", range(10))
def test_c_oracle_designs_on_streams(idx):
  g = load_golden("timevar.json")
  c = g["designs"][idx]
  assert same_bits(oracle.tv_df1_c(as_arrays(coefs(c["b"])), as_arrays(coefs(c["a"])), unhex(g["x"])), unhex(c["y"]))


def test_c_oracle_equals_the_python_form_on_random_banks():
  """Shared and per-channel series, every tap pattern of a biquad, gains, memory and zero: the C form against
  oracle.tv_df1 (itself pinned on the reference's vectors) on every channel."""
  rng = np.random.default_rng(77)
  N, C = 300, 5
  for trial in range(40):
    x = rng.uniform(-1, 1, (N, C))
    def tap(allow_zero=True):
      r = rng.integers(0, 4)
      if r == 0 and allow_zero:
        return 0.
      if r == 1:
        return float(rng.uniform(-.9, .9))
      if r == 2:
        return rng.uniform(-.9, .9, N)
      return rng.uniform(-.9, .9, (N, C))
    b = [tap() for _ in range(rng.integers(1, 4))]
    a = [float(rng.choice([1., -1., 2., .5]))] + [tap() for _ in range(rng.integers(0, 3))]
    zero = float(rng.choice([0., .25]))
    memory = None if rng.random() < .5 else rng.uniform(-1, 1, len(a) - 1).tolist()
    layout = "time" if trial % 2 == 0 else "chan"
    xin = x if layout == "time" else np.ascontiguousarray(x.T)
    got = oracle.tv_bank(b, a, xin, layout=layout, memory=memory, zero=zero)
    for c in range(C):
      bc = [v[:, c] if isinstance(v, np.ndarray) and v.ndim == 2 else v for v in b]
      ac = [v[:, c] if isinstance(v, np.ndarray) and v.ndim == 2 else v for v in a]
      ref = oracle.tv_df1(bc, ac, x[:, c], memory=memory, zero=zero)
      assert same_bits(got[:, c], ref), (trial, c)
