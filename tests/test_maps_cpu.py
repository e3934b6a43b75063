"""The elementwise stages of SURVEY.md 8 (f1) on the CPU side: the oracle's NumPy restatement
(oracle.map_block) and the host-side mirrors (clip, maverage.deque, amdf) against vectors produced
by the reference itself (tests/golden/maps.json, oracle/gen_golden.py::maps_cases)."""
import numpy as np
import pytest

from conftest import load_golden, unhex
from oracle import oracle

G = load_golden("maps.json")
X, Y, XS = unhex(G["x"]), unhex(G["y"]), unhex(G["xs"])


def same_bits(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return a.shape == b.shape and bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


def test_oracle_unary_and_scalar_ops_bit_exact():
  for case in G["unary"]:
    x = unhex(case["x"]) if "x" in case else X
    if case["op"] == "square_pow":
      continue
    assert same_bits(oracle.map_block(case["op"], x), unhex(case["r"])), case["op"]
  for case in G["scalar"]:
    src = Y if case["op"] == "rdiv" else X
    op = "mul" if case["op"] == "rmul" else case["op"]
    assert same_bits(oracle.map_block(op, src, p0=unhex(case["c"])), unhex(case["r"])), case["op"]
  for case in G["binary"]:
    assert same_bits(oracle.map_block(case["op"] + "2", X, Y), unhex(case["r"])), case["op"]


def test_oracle_clip_rules_bit_exact():
  for case in G["clip"]:
    low = None if case["low"] is None else unhex(case["low"])
    high = None if case["high"] is None else unhex(case["high"])
    if low is None and high is None:
      got = np.array(X)
    elif low is None:
      got = oracle.map_block("clip_high", X, p1=high)
    elif high is None:
      got = oracle.map_block("clip_low", X, p0=low)
    else:
      got = oracle.map_block("clip", X, p0=low, p1=high)
    assert same_bits(got, unhex(case["r"])), (low, high)


def test_x_times_x_is_not_the_references_power():
  # the reason "square" is opt-in: libm's pow(x, 2.0) is not the correctly rounded product everywhere
  case = [c for c in G["unary"] if c["op"] == "square_pow"][0]
  x, ref = np.array(unhex(case["x"])), np.array(unhex(case["r"]))
  prod = oracle.map_block("square", x)
  assert np.max(np.abs(prod - ref) / np.maximum(np.abs(ref), 1e-300)) <= 2.3e-16     # one ulp at most
  # (whether any of these 300 samples differs depends on the libm the fixtures were made with)


def test_host_clip_mirrors_the_reference():
  import audiolazy_amd as alz
  for case in G["clip"]:
    low = None if case["low"] is None else unhex(case["low"])
    high = None if case["high"] is None else unhex(case["high"])
    assert same_bits(list(alz.clip(X, low, high)), unhex(case["r"])), (low, high)
  with pytest.raises(ValueError):
    alz.clip(X, 1., -1.)


def test_host_running_mean_mirrors_the_reference():
  import audiolazy_amd as alz
  for case in G["callers"]:
    if case["fn"] == "maverage.deque":
      got = list(alz.maverage.deque(case["size"])(XS, zero=unhex(case["zero"])))
      assert same_bits(got, unhex(case["r"])), case


def test_root_edge_items_follow_python_pow():
  """``v ** .5`` in CPython is libm pow(v, .5): (-0.0) ** .5 is +0.0 and (-inf) ** .5 is +inf, where a
  plain square root gives -0.0 and NaN.  The restatement follows the per-sample expression."""
  from oracle import oracle
  items = [0.0, -0.0, 4.0, 2.0, float("inf"), float("-inf"), float("nan"), 5e-324, 1.7976931348623157e308]
  got = oracle.map_block("sqrt", np.array(items))
  want = np.array([v ** .5 for v in items])
  assert np.array_equal(got.view(np.uint64)[:6], want.view(np.uint64)[:6])
  assert np.isnan(got[6]) and np.isnan(want[6])
  assert np.array_equal(got.view(np.uint64)[7:], want.view(np.uint64)[7:])
