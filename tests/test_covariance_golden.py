"""The covariance-method callers beside the LPC path -- ``lag_matrix`` (reference lazy_analysis.py:315-342),
``lpc.covar`` / ``lpc.kcovar`` (lazy_lpc.py:275-340), ``parcor`` / ``parcor_stable`` (:343-425), ``toeplitz``
(:44-49) -- against tests/golden/covariance.json, which oracle/gen_golden.py --only-covariance wrote from the
unmodified reference.  CPU part: the C restatement of lag_matrix bit for bit; the host algebra (the greedy kcovar
solve, the pinv solve, the backwards Levinson-Durbin) with the oracle's lag matrix standing in for the engine's;
integer blocks, which never reach the engine.  tests/test_gpu_covariance.py runs the same cases through the kernel."""
import ast
import importlib

import numpy as np
import pytest

from conftest import load_golden

G = load_golden("covariance.json")


def unhex(v):
  return [unhex(i) for i in v] if isinstance(v, list) else float.fromhex(v)


def block(name):
  raw = G["blocks"][name]
  return ast.literal_eval(raw) if isinstance(raw, str) else unhex(raw)


def same_bits(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def lpc_module():
  return importlib.import_module("audiolazy_amd.lpc")     # (the package attribute `lpc` is the StrategyDict)


def filt_outcome(fn):
  try:
    f = fn()
  except Exception as exc:   # noqa: BLE001 -- type and text are what is compared
    return dict(raises=type(exc).__name__, text=str(exc))
  return dict(value=dict(coefs=[float(v).hex() for v in f.numlist], first_is_int=isinstance(f.numlist[0], int),
                         den=[float(v).hex() for v in f.denlist], error=float(f.error).hex(),
                         error_type=type(f.error).__name__))


def expected(case):
  return {k: case[k] for k in ("value", "raises", "text") if k in case}


FLOAT_LAGS = [c for c in G["lag_matrix"] if "phi" in c and c["blk"] != "mixed"]


@pytest.mark.parametrize("case", FLOAT_LAGS, ids=lambda c: "%s-%s" % (c["blk"], c["max_lag"]))
def test_oracle_lag_matrix_is_the_reference(case):
  from oracle import oracle
  assert same_bits(oracle.lag_matrix(block(case["blk"]), case["max_lag"]), unhex(case["phi"]))


def test_oracle_lag_matrix_refuses_what_the_reference_refuses():
  from oracle import oracle
  for case in G["lag_matrix"]:
    if "raises" in case:
      with pytest.raises(ValueError, match="Block length should be higher than order"):
        oracle.lag_matrix(block(case["blk"]), case["max_lag"])


def test_integer_blocks_stay_integer_on_the_host():
  """A block of nothing but ints is integer arithmetic in the reference: exact, and ints come back."""
  import audiolazy_amd as al
  for case in G["lag_matrix"]:
    if "phi_repr" in case:
      got = al.lag_matrix(block("ints"), case["max_lag"])
      assert repr(got) == case["phi_repr"]
  big = [2 ** 40 + 1, -3, 2 ** 35]
  assert al.lag_matrix(big, 1) == [[sum(big[n] * big[n] for n in (1, 2)), sum(big[n - 1] * big[n] for n in (1, 2))],
                                   [sum(big[n] * big[n - 1] for n in (1, 2)), sum(big[n - 1] * big[n - 1] for n in (1, 2))]]
  assert al.acorr([1, 2, 3, 4]) == [30, 20, 11, 4] and all(isinstance(v, int) for v in al.acorr([1, 2, 3, 4]))
  assert al.acorr([], 2) == [0, 0, 0] and al.acorr([]) == [] and al.lag_matrix([]) == []


def test_lag_matrix_argument_checks_need_no_engine():
  import audiolazy_amd as al
  for case in G["lag_matrix"]:
    if "raises" in case:
      with pytest.raises(ValueError, match=case["text"]):
        al.lag_matrix(block(case["blk"]), case["max_lag"])
  with pytest.raises(ValueError, match="Block length should be higher than order"):
    al.lag_matrix_frames(np.zeros(64), 8, 8)


def test_toeplitz():
  import audiolazy_amd as al
  for case in G["toeplitz"]:
    assert repr(al.toeplitz(ast.literal_eval(case["vect"]))) == case["matrix"]


@pytest.fixture
def oracle_lags(monkeypatch):
  """The host algebra with the oracle's lag matrix where the engine's would be (no GPU here)."""
  from oracle import oracle
  mod = lpc_module()

  def frames(sig, frame_len, max_lag, hop=None, device=0):
    sig = np.asarray(sig, dtype=np.float64).reshape(-1)
    assert len(sig) == frame_len
    return oracle.lag_matrix(sig, max_lag)[None]
  monkeypatch.setattr(mod, "lag_matrix_frames", frames)
  monkeypatch.setattr(mod, "acorr_frames", lambda sig, frame_len, max_lag, hop=None, device=0:
                      oracle.acorr(np.asarray(sig, dtype=np.float64), max_lag)[None])
  return mod


def test_cells_without_a_float_term_stay_integers(oracle_lags):
  """In a block that mixes ints and floats the reference's sums start from int 0: a cell (or lag) whose terms are
  all int x int comes back as an int, and so does the empty sum of a lag past the block."""
  import audiolazy_amd as al
  blk = [-4, 0, -2, 4, 4, -3, -3, 1, 0.2493200140664249, 4, 1.8972099364840025, -4, 0.3933814750538396]
  for lag in (None, 2, 5):
    m = len(blk) - 1 if lag is None else lag
    want = [[sum(blk[n - i] * blk[n - j] for n in range(m, len(blk))) for i in range(m + 1)] for j in range(m + 1)]
    got = al.lag_matrix(blk, lag)
    assert repr(got) == repr(want)
  assert any(isinstance(v, int) for row in al.lag_matrix(blk) for v in row)
  for lag in (None, 3, len(blk) + 2):
    m = len(blk) - 1 if lag is None else lag
    want = [sum(blk[n] * blk[n + tau] for n in range(len(blk) - tau)) for tau in range(m + 1)]
    assert repr(al.acorr(blk, lag)) == repr(want)
  assert repr(al.acorr([.5, .25], 4)) == repr([.3125, .125, 0, 0, 0])
  huge = [2 ** 30, .5, 2 ** 30]                          # products past 2**53: not the engine's block
  assert repr(al.acorr(huge)) == repr([sum(huge[n] * huge[n + t] for n in range(3 - t)) for t in range(3)])


@pytest.mark.parametrize("family", ["covar", "kcovar"])
def test_covariance_solves_on_the_host(oracle_lags, family):
  import audiolazy_amd as al
  for case in G[family]:
    strategy = al.lpc[case.get("alias", family)]
    got = filt_outcome(lambda: strategy(block(case["blk"]), case["order"]))
    assert got == expected(case), (family, case["blk"], case["order"], case.get("alias"))


def test_covariance_aliases_are_the_reference_s():
  import audiolazy_amd as al
  for alias in ("covar", "cov", "covariance", "ncovar", "ncov", "ncovariance"):
    assert al.lpc[alias] is al.lpc.covar
  for alias in ("kcovar", "kcov", "kcovariance"):
    assert al.lpc[alias] is al.lpc.kcovar
  assert al.lpc.default is al.lpc.autocor


def test_parcor_runs_levinson_durbin_backwards():
  import audiolazy_amd as al
  for case in G["parcor"]:
    filt = al.ZFilter(unhex(case["num"]), unhex(case["den"]))
    if case["name"] == "ints":
      filt = al.ZFilter([1, 2, 3])
    got, err = [], None
    try:
      for k in al.parcor(filt):
        got.append(k)
    except Exception as exc:   # noqa: BLE001
      err = (type(exc).__name__, str(exc))
    assert [float(k).hex() for k in got] == case["ks"], case["name"]
    assert err == ((case["raises"], case["text"]) if "raises" in case else None), case["name"]
  gen = al.parcor(al.ZFilter([1., .5], [1., .2]))       # a generator: the refusal comes with the first next()
  with pytest.raises(ValueError, match="Filter has feedback"):
    next(gen)
  assert issubclass(al.ParCorError, ZeroDivisionError)


def test_parcor_stable():
  import audiolazy_amd as al
  for case in G["parcor_stable"]:
    filt = al.ZFilter(unhex(case["num"]), unhex(case["den"]))
    assert al.parcor_stable(filt) is case["value"], case["name"]


def test_lsf_alternates_for_a_minimum_phase_filter():
  """The reference's lsf cannot run under NumPy 2 (its elementwise phase builds an np.mat), so no generated golden
  (tests/test_reference_lpc.py holds it to the hand-checked vectors of the reference's tests); here the properties it
  documents -- angles of P and Q interleave, starting with the lowest, for a stable filter;
  0 and pi are the trivial roots; an unstable denominator breaks the alternation."""
  import math
  import audiolazy_amd as al
  fir = al.ZFilter([1., -1.2, .8, -.3])                 # zeros inside the unit circle
  assert al.parcor_stable(1 / fir)
  freqs = al.lsf(fir)
  assert len(freqs) == 8 and all(a < b for a, b in zip(freqs, freqs[1:]))
  assert al.lsf_stable(1 / fir)
  assert any(abs(f) < 1e-12 for f in freqs) and any(abs(abs(f) - math.pi) < 1e-12 for f in freqs)
  # the angles are roots of P and Q: A(w) +- e^{-jw(p+1)} conj(A(w)) vanishes there
  for w in freqs:
    e = complex(math.cos(w), -math.sin(w))
    a = sum(c * e ** k for k, c in enumerate(fir.numlist))
    a_rev = sum(c * e ** (len(fir.numlist) - k) for k, c in enumerate(fir.numlist))
    assert min(abs(a + a_rev), abs(a - a_rev)) < 1e-9
  assert not al.lsf_stable(1 / al.ZFilter([1., -2.5, 1.]))
  with pytest.raises(ValueError, match="Filter has feedback"):
    al.lsf(al.ZFilter([1., .5], [1., .2]))
