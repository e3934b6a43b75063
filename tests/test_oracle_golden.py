"""Pin the CPU oracle (oracle/alz_oracle.c) against the reference.

The golden vectors under tests/golden/ were produced by running AudioLazy
itself (oracle/gen_golden.py); the literal known answers below are re-stated
from the reference's doctests and tests, cited per case.  Integer/byte work
would be bit-exact; here the arithmetic is float64 and the DF-I restatement is
*still* required to be bit-identical (np.array_equal on the raw doubles,
tolerance 0) because it follows the generated expression term by term.
The Levinson restatement is held to 1e-12 relative (it follows the same sums,
bit-identity is checked where it holds and reported otherwise).
"""
import numpy as np
import pytest

from conftest import load_golden, unhex
from oracle import oracle


def bits(a):
  return np.asarray(a, dtype=np.float64).view(np.uint64)


def same_bits(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  # NaN payloads aside, every double must match exactly (signed zeros included)
  return a.shape == b.shape and bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


FILT = load_golden("filters.json")


@pytest.mark.parametrize("case", FILT["cases"], ids=lambda c: c["tag"])
def test_df1_bit_exact_vs_reference(case):
  base = unhex(FILT["x"])
  x = unhex(case["x"]) if "x" in case else base[:case["x_len"]]
  mem = None if case["memory"] is None else unhex(case["memory"])
  y = oracle.df1(unhex(case["b"]), unhex(case["a"]), x, memory=mem, zero=unhex(case["zero"]))
  assert same_bits(y, unhex(case["y"]))


def test_lfilter_grid_bit_exact():
  # reference tests/test_filters_extdep.py:41-47 (cases run on the reference)
  for case in load_golden("lfilter_grid.json"):
    y = oracle.df1(unhex(case["b"]), unhex(case["a"]), unhex(case["x"]))
    assert same_bits(y, unhex(case["y"]))


def test_lfilter_grid_vs_scipy():
  # the reference's own oracle for LTI filters is scipy.signal.lfilter (README.rst:219-220)
  scipy_signal = pytest.importorskip("scipy.signal")
  for case in load_golden("lfilter_grid.json"):
    b, a, x = unhex(case["b"]), unhex(case["a"]), unhex(case["x"])
    y = oracle.df1(b, a, x)
    np.testing.assert_allclose(y, scipy_signal.lfilter(b, a, x), rtol=2 ** -23, atol=1e-12)


def test_known_answers_from_reference_docs():
  # README.rst:286-300: (1 - z**-1) on [.1,.2,.4,.3,.2,-.1,-.3,-.2]
  data = [.1, .2, .4, .3, .2, -.1, -.3, -.2]
  y = oracle.df1([1., -1.], [1.], data)
  np.testing.assert_allclose(y, [.1, .1, .2, -.1, -.1, -.3, -.2, .1], rtol=0, atol=1e-15)
  # audiolazy/__init__.py:30-33 accumulator 1/(1 - z**-1)
  y = oracle.df1([1.], [1., -1.], [1, 3, 2, 1, 3, 2, 1, 3])
  assert list(y) == [1, 4, 6, 7, 10, 12, 13, 16]
  # lazy_filters.py:722-726: (1 + z**-1)/(1 - z**-1)
  assert list(oracle.df1([1., 1.], [1., -1.], [1, 5, -4, -7, 9])) == [1, 7, 8, -3, -1]
  # lazy_filters.py:735-742: same filter, memory=[3], zero=0 -> [4, 10, 11, 0, 2]
  assert list(oracle.df1([1., 1.], [1., -1.], [1, 5, -4, -7, 9], memory=[3], zero=0)) == [4, 10, 11, 0, 2]
  # tests/test_filters.py:161-169  y[n] = x[n-1] - y[n-2]
  x = [1., 2., 3., 4., 5., 6.]
  exp = []
  for n in range(6):
    exp.append((x[n - 1] if n >= 1 else 0.) - (exp[n - 2] if n >= 2 else 0.))
  assert list(oracle.df1([0., 1.], [1., 0., 1.], x)) == exp


def test_zero_gain_raises():
  with pytest.raises(ZeroDivisionError):  # lazy_filters.py:177-178
    oracle.df1([1.], [0., 1.], [1., 2.])


def test_all_zero_filter_yields_zero():
  # lazy_filters.py:227-231
  assert list(oracle.df1([0.], [1.], [1., 2., 3.], zero=0.5)) == [.5, .5, .5]


def test_memory_left_pad_quirk():
  # lazy_filters.py:193-195 + lazy_misc.py:132 (left padding)
  assert oracle.memory_to_hist([.7], 2, 0.) == [0., .7]
  assert oracle.memory_to_hist([1, 2, 3], 2, 0.) == [1, 2]
  assert oracle.memory_to_hist(None, 3, .5) == [.5, .5, .5]
  assert oracle.memory_to_hist(lambda n: [9.] * n, 2, 0.) == [9., 9.]


def test_block_continuity_equals_one_run():
  base = unhex(FILT["x"])
  b, a = [0.2, 0.3, 0.4], [2., -.5, .25]
  whole = oracle.df1(b, a, base)
  st = {}
  parts = [oracle.df1(b, a, base[i:i + 37], state=st) for i in range(0, len(base), 37)]
  assert same_bits(np.concatenate(parts), whole)


@pytest.mark.parametrize("idx", [0, 1])
def test_cascade_bit_exact(idx):
  case = load_golden("containers.json")[idx]
  assert case["kind"] == "cascade"
  secs = case["sections"]
  nb = [len(s["b"]) for s in secs]
  na = [len(s["a"]) for s in secs]
  bc = np.concatenate([unhex(s["b"]) for s in secs])
  ac = np.concatenate([unhex(s["a"]) for s in secs])
  x = np.array(unhex(case["x"]))
  zero = unhex(case["zero"])
  xh = np.full((1, sum(n - 1 for n in nb)), zero)
  mem = None if case["memory"] is None else unhex(case["memory"])
  yh = np.concatenate([oracle.memory_to_hist(mem, n - 1, zero) for n in na]).reshape(1, -1)
  y = oracle.bank(nb, na, bc, ac, x.reshape(-1, 1), layout="time", xh=xh, yh=np.ascontiguousarray(yh, dtype=float), zero=zero)
  assert same_bits(y[:, 0], unhex(case["y"]))


def test_parallel_bit_exact():
  case = load_golden("containers.json")[2]
  x = unhex(case["x"])
  ys = [oracle.df1(unhex(s["b"]), unhex(s["a"]), x) for s in case["sections"]]
  tot = ys[0]
  for y in ys[1:]:
    tot = tot + y  # ((f1 + f2) + f3), lazy_filters.py:1052-1054
  assert same_bits(tot, unhex(case["y"]))


def test_gammatone_cascades_bit_exact():
  aud = load_golden("auditory.json")
  x = np.array(unhex(aud["x"]))
  for g in aud["gammatone"]:
    secs = g["sections"]
    assert len(secs) == 4 and all(len(s["a"]) == 3 for s in secs)  # tests/test_auditory.py:78-92
    nb = [len(s["b"]) for s in secs]
    na = [3] * 4
    y = oracle.bank(nb, na, np.concatenate([unhex(s["b"]) for s in secs]),
                    np.concatenate([unhex(s["a"]) for s in secs]), x.reshape(-1, 1))
    assert same_bits(y[:, 0], unhex(g["y"])), g["strategy"]


def test_multichannel_both_layouts_bit_exact():
  mc = load_golden("multichannel.json")
  b, a = np.array(unhex(mc["b"])), np.array(unhex(mc["a"]))
  x, y = np.array(unhex(mc["x"])), np.array(unhex(mc["y"]))
  assert same_bits(oracle.bank([3], [3], b, a, x, layout="time"), y)
  assert same_bits(oracle.bank([3], [3], b, a, np.ascontiguousarray(x.T), layout="chan"), y.T)


def test_acorr_bit_exact():
  for case in load_golden("lpc.json")["acorr"]:
    x = unhex(case["x"])
    lag = case["max_lag"]
    assert same_bits(oracle.acorr(x, lag), unhex(case["r"]))
  # lazy_analysis.py:298-306
  assert list(oracle.acorr([1, 2, 3, 4, 3, 4, 2])) == [59, 52, 42, 30, 17, 8, 2]


def test_levinson_vs_reference():
  worst = 0.0
  for case in load_golden("lpc.json")["levinson"]:
    ac = unhex(case["ac"])
    coefs, err = oracle.levinson_durbin(ac, case["order"])
    ref = np.array(unhex(case["coefs"]))
    ref = np.concatenate([ref, np.zeros(len(coefs) - len(ref))])
    np.testing.assert_allclose(coefs, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    assert err == pytest.approx(unhex(case["error"]), rel=1e-12, abs=1e-12)
    worst = max(worst, np.abs(coefs - ref).max())
  # lazy_lpc.py:99-107
  coefs, err = oracle.levinson_durbin([12, 6, 0, -3, -6, -3, 0, 2, 4, 2], 3)
  assert list(coefs) == [1, -0.625, 0.25, 0.125] and err == 7.875
  # tests/test_lpc.py:280-290  levinson_durbin([1, 5, 3])
  coefs, err = oracle.levinson_durbin([1., 5., 3.])
  np.testing.assert_allclose(coefs, [1, -5. / 12, -11. / 12], rtol=1e-14)


def test_kautocor_vs_reference():
  for case in load_golden("lpc.json")["kautocor"]:
    x = unhex(case["x"])
    coefs, err, status = oracle.kautocor_frames(x, 1, len(x), len(x), case["order"])
    ref = np.array(unhex(case["coefs"]))
    ref = np.concatenate([ref, np.zeros(coefs.shape[1] - len(ref))])
    assert status[0] == 0
    np.testing.assert_allclose(coefs[0], ref, rtol=1e-12, atol=1e-12)
    assert err[0] == pytest.approx(unhex(case["error"]), rel=1e-11, abs=1e-12)
  # tests/test_lpc.py:218-224: [-1,0,1,0]*4 order 2 -> 1 + 0.875 z^-2, error 1.875
  c, e, st = oracle.kautocor_frames([-1., 0., 1., 0.] * 4, 1, 16, 16, 2)
  np.testing.assert_allclose(c[0], [1, 0, .875], atol=1e-15)
  assert e[0] == pytest.approx(1.875, rel=1e-14)


def test_kautocor_zero_frame_is_parcor_error():
  # lazy_lpc.py:132-133
  c, e, st = oracle.kautocor_frames(np.zeros(32), 1, 32, 32, 4)
  assert st[0] == -4


# ---------------------------------------------------------------------------
# oracle/pyref.py: the generated CPython generator restated (bench.py's reference-path CPU legs)
# ---------------------------------------------------------------------------
def test_pyref_generated_loop_bit_exact_vs_reference():
  from oracle import pyref
  base = unhex(FILT["x"])
  for case in FILT["cases"]:
    x = unhex(case["x"]) if "x" in case else base[:case["x_len"]]
    mem = None if case["memory"] is None else unhex(case["memory"])
    y = list(pyref.df1(unhex(case["b"]), unhex(case["a"]), x, memory=mem, zero=unhex(case["zero"])))
    assert same_bits(y, unhex(case["y"])), case["tag"]


def test_pyref_series_coefficients_bit_exact_vs_reference():
  from oracle import pyref
  tv = load_golden("timevar.json")
  x = unhex(tv["x"])

  def coef(c):
    return unhex(c["series"]) if "series" in c else unhex(c["const"])
  for case in tv["direct"]:
    b, a = [coef(c) for c in case["b"]], [coef(c) for c in case["a"]]
    if hasattr(a[0], "__iter__"):
      continue          # a series a0 is normalised away by the filter algebra before the loop exists
    mem = None if case["memory"] is None else unhex(case["memory"])
    y = list(pyref.df1(b, a, x, memory=mem, zero=unhex(case["zero"])))
    ref = unhex(case["y"])
    assert same_bits(y[:len(ref)], ref)


def test_pyref_source_is_the_documented_statement():
  # SURVEY.md appendix A (hooked _exec_eval): resonator.z_exp(1000 Hz, 100 Hz)
  from oracle import pyref
  src, names = pyref.df1_source([0.0065023341711623606, 0.0, -0.0065023341711623606],
                                [1, -1.9699963111457524, 0.9869953316576753])
  assert names == []
  assert ("m0 = 0.0065023341711623606 * d0 + -0.0065023341711623606 * d2 + --1.9699963111457524 * m1 "
          "+ -0.9869953316576753 * m2") in src
  assert "m1 , m2 , = memory" in src and "d1 = d2 = zero" in src
  src, names = pyref.df1_source([[.1, .2]], [1, [.3, .4]])
  assert names == ["b0", "a1"] and "m0 = next(b0) * d0 + -next(a1) * m1" in src
  assert pyref.consume_blocks(pyref.noise(10000, 1), 4096) == 10000


def test_lpc_strategies_against_the_reference():
  """tests/golden/lpc_strategies.json (generated by the reference): lags past 63 and Levinson-Durbin at orders
  40 .. 120 are bit-identical in the C restatement; the pseudo-inverse strategies go through numpy.linalg.pinv
  like the reference (LAPACK results may differ in the last bits from one CPU to another: 1e-9)."""
  g = load_golden("lpc_strategies.json")
  blocks = {k: unhex(v) for k, v in g["blocks"].items()}
  for case in g["acorr"]:
    assert same_bits(oracle.acorr(blocks[case["blk"]], case["max_lag"]), unhex(case["r"]))
  for case in g["kautocor"] + [c for c in g["autocor"] if c["route"] == "kautocor"]:
    x = blocks[case["blk"]]
    coefs, err, status = oracle.kautocor_frames(x, 1, len(x), len(x), case["order"])
    ref = np.array(unhex(case["coefs"]))
    ref = np.concatenate([ref, np.zeros(coefs.shape[1] - len(ref))])
    assert status[0] == 0 and same_bits(coefs[0], ref) and err[0] == unhex(case["error"])
  for case in g["nautocor"] + [c for c in g["autocor"] if c["route"] != "kautocor"]:
    x = blocks[case["blk"]]
    coefs, err = oracle.nautocor(x, case["order"])
    ref = unhex(case["coefs"])
    ref = ref + [0.] * (len(coefs) - len(ref))             # Poly drops exact-zero top coefficients
    np.testing.assert_allclose(coefs, ref, rtol=1e-9, atol=1e-9)
    assert err == pytest.approx(unhex(case["error"]), rel=1e-9, abs=1e-9)
  # the silent block: kautocor is a ParCorError (status), the fallback's coefficients are all zero
  assert oracle.kautocor_frames(blocks["silent"], 1, 150, 150, 100)[2][0] == -4


def test_levinson_and_kautocor_restatements_are_bit_identical():
  """Tighter than the 1e-12 of the round-1 tests: the C restatement of the dense Levinson-Durbin reproduces the
  reference's coefficients and error to the last bit on every golden case."""
  for case in load_golden("lpc.json")["levinson"]:
    coefs, err = oracle.levinson_durbin(unhex(case["ac"]), case["order"])
    ref = np.array(unhex(case["coefs"]))
    ref = np.concatenate([ref, np.zeros(len(coefs) - len(ref))])
    assert same_bits(np.asarray(coefs, dtype=np.float64), ref) and err == unhex(case["error"])
  for case in load_golden("lpc.json")["kautocor"]:
    x = unhex(case["x"])
    coefs, err, status = oracle.kautocor_frames(x, 1, len(x), len(x), case["order"])
    ref = np.array(unhex(case["coefs"]))
    ref = np.concatenate([ref, np.zeros(coefs.shape[1] - len(ref))])
    assert status[0] == 0 and same_bits(coefs[0], ref) and err[0] == unhex(case["error"])
