"""GPU parity for batched lpc.kautocor (reference lazy_lpc.py:229-272).

acorr is bit-exact (same left-to-right sum).  Levinson-Durbin on the GPU is by default the
standard O(order^2) recursion, not the reference's dense inner products, so it
is floating-point parity: normalised max error of the coefficient vector and
relative error of .error must be <= 1e-9 (north star allows 1e-6).  With exact=True
(ALZ_LPC_DENSE) the reference's dense form runs and everything is bit-identical (tests at the end).
"""
import numpy as np
import pytest

from conftest import load_golden, unhex

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(scope="module")
def lpc():
  import audiolazy_amd
  assert audiolazy_amd.device_count() >= 1
  import importlib
  return importlib.import_module("audiolazy_amd.lpc")   # (audiolazy_amd.lpc the attribute is the StrategyDict)


def test_golden_acorr_bit_exact(lpc):
  for case in load_golden("lpc.json")["acorr"]:
    x = np.array(unhex(case["x"]))
    lag = case["max_lag"] if case["max_lag"] is not None else len(x) - 1
    if lag > 63:
      continue
    r = lpc.acorr_frames(x, len(x), lag)
    assert np.array_equal(r[0].view(np.uint64), np.array(unhex(case["r"])).view(np.uint64))


def test_golden_kautocor(lpc):
  for case in load_golden("lpc.json")["kautocor"]:
    x = np.array(unhex(case["x"]))
    ref = np.array(unhex(case["coefs"]))
    c, e, st = lpc.kautocor_frames(x, len(x), case["order"])
    ref = np.concatenate([ref, np.zeros(c.shape[1] - len(ref))])
    assert st[0] == 0
    assert np.abs(c[0] - ref).max() / np.abs(ref).max() <= TOL
    assert abs(e[0] - unhex(case["error"])) <= TOL * max(abs(unhex(case["error"])), 1e-300) + 1e-12


def test_known_answers(lpc):
  # reference tests/test_lpc.py:218-224: [-1,0,1,0]*4, order 2 -> 1 + 0.875 z^-2, error 1.875
  c, e, st = lpc.kautocor_frames(np.array([-1., 0., 1., 0.] * 4), 16, 2)
  np.testing.assert_allclose(c[0], [1, 0, .875], atol=1e-14)
  assert e[0] == pytest.approx(1.875, rel=1e-13)
  # reference lazy_lpc.py:250-259 style: the doctest signal of levinson_durbin, through kautocor
  data = np.array([2., 2, 0, 0, -1, -1, 0, 0, 1, 1])
  c, e, st = lpc.kautocor_frames(data, 10, 3)
  np.testing.assert_allclose(c[0], [1, -0.625, 0.25, 0.125], atol=1e-14)
  assert e[0] == pytest.approx(7.875, rel=1e-13)


def test_batch_vs_oracle_with_hop_and_zero_frames(lpc):
  from oracle import oracle
  rng = np.random.default_rng(99)
  L, hop, order = 480, 240, 16
  F = 1000
  sig = rng.uniform(-1, 1, (F - 1) * hop + L)
  sig[20 * hop: 20 * hop + L] = 0.0   # frame 20 has zero energy -> ParCorError in the reference
  c, e, st = lpc.kautocor_frames(sig, L, order, hop=hop)
  rc, re, rst = oracle.kautocor_frames(sig, F, L, hop, order)
  assert c.shape == (F, order + 1)
  assert st[20] == -4 and rst[20] == -4        # ALZ_E_PARCOR
  ok = rst == 0
  assert np.array_equal(st == 0, ok)
  err = np.abs(c[ok] - rc[ok]).max(axis=1) / np.abs(rc[ok]).max(axis=1)
  assert err.max() <= TOL
  assert (np.abs(e[ok] - re[ok]) / np.abs(re[ok])).max() <= TOL


def test_order_above_31_uses_64_lane_slots(lpc):
  from oracle import oracle
  rng = np.random.default_rng(5)
  L, order, F = 300, 40, 17
  sig = rng.uniform(-1, 1, F * L)
  c, e, st = lpc.kautocor_frames(sig, L, order)
  rc, re, rst = oracle.kautocor_frames(sig, F, L, L, order)
  assert (st == 0).all()
  assert (np.abs(c - rc).max(axis=1) / np.abs(rc).max(axis=1)).max() <= 1e-7


def test_reference_operator_surface(lpc):
  """acorr / levinson_durbin / lpc.kautocor as single-block callables returning what the
  reference returns (a list; a ZFilter with .error)."""
  import audiolazy_amd as alz
  assert alz.acorr([1, 2, 3, 4, 3, 4, 2]) == [59, 52, 42, 30, 17, 8, 2]      # lazy_analysis.py:298-306
  assert alz.acorr([1, 2, 3, 4, 3, 4, 2], 9)[-3:] == [0, 0, 0]
  f = alz.levinson_durbin([12, 6, 0, -3, -6, -3, 0, 2, 4, 2], 3)              # lazy_lpc.py:99-107
  np.testing.assert_allclose(f.numlist, [1, -0.625, 0.25, 0.125], atol=1e-14)
  assert f.error == pytest.approx(7.875, rel=1e-13) and f.denlist == [1]
  f = alz.levinson_durbin([1., 5., 3.])                                       # tests/test_lpc.py:280-290
  np.testing.assert_allclose(f.numlist, [1, -5. / 12, -11. / 12], rtol=1e-13)
  from oracle import oracle
  f = alz.levinson_durbin([1., .5], 3)                                        # zero-extended lags (:117-118)
  np.testing.assert_allclose(f.numlist, oracle.levinson_durbin([1., .5], 3)[0], atol=1e-14)
  for case in load_golden("lpc.json")["levinson"]:
    f = alz.levinson_durbin(unhex(case["ac"]), case["order"])
    ref = unhex(case["coefs"])
    np.testing.assert_allclose(f.numlist + [0.] * (len(ref) - len(f.numlist)), ref, rtol=1e-9, atol=1e-9)
    assert f.error == pytest.approx(unhex(case["error"]), rel=1e-9, abs=1e-9)
  g = alz.lpc.kautocor([-1., 0., 1., 0.] * 4, 2)                             # tests/test_lpc.py:218-224
  np.testing.assert_allclose(g.numlist, [1, 0, .875], atol=1e-14)
  assert g.error == pytest.approx(1.875, rel=1e-13)
  assert alz.lpc([-1., 0., 1., 0.] * 4, 2).numlist == g.numlist
  with pytest.raises(alz.ParCorError):                                        # lazy_lpc.py:132-133
    alz.lpc.kautocor([0.] * 32, 4)
  with pytest.raises(ZeroDivisionError):
    alz.levinson_durbin([0., 0., 0.])


def test_large_batch_lane_kernels_bit_exact_acorr(lpc):
  """>= 16384 frames take the lane-per-frame kernels (k_acorr_lane / k_levinson_lane): acorr must
  still be bit-identical to the oracle's left-to-right sums, including overlapping frames."""
  from oracle import oracle
  rng = np.random.default_rng(123)
  L, hop, order, F = 480, 240, 16, 20000
  sig = rng.uniform(-1, 1, (F - 1) * hop + L)
  sig[5 * hop: 5 * hop + L] = 0.0
  r = lpc.acorr_frames(sig, L, order, hop=hop)
  pick = [0, 1, 5, 777, F - 1]
  for f in pick:
    assert np.array_equal(r[f].view(np.uint64), oracle.acorr(sig[f * hop: f * hop + L], order).view(np.uint64))
  c, e, st = lpc.kautocor_frames(sig, L, order, hop=hop)
  rc, re, rst = oracle.kautocor_frames(sig, F, L, hop, order)
  assert st[5] == -4 and rst[5] == -4
  ok = rst == 0
  assert np.array_equal(st == 0, ok)
  assert (np.abs(c[ok] - rc[ok]).max(axis=1) / np.abs(rc[ok]).max(axis=1)).max() <= TOL
  assert (np.abs(e[ok] - re[ok]) / np.abs(re[ok])).max() <= TOL
  # a short frame length (tail loop only) and an odd order that has no lane instantiation
  sig2 = rng.uniform(-1, 1, 17000 * 40)
  r2 = lpc.acorr_frames(sig2, 40, 16)
  assert np.array_equal(r2[16999].view(np.uint64), oracle.acorr(sig2[16999 * 40:], 16).view(np.uint64))
  r3 = lpc.acorr_frames(sig2, 40, 7)
  assert np.array_equal(r3[3].view(np.uint64), oracle.acorr(sig2[120:160], 7).view(np.uint64))


def test_fused_mode_within_contract(lpc):
  """ALZ_LPC_FUSED (opt-in): fused multiply-adds in the autocorrelation sums; the coefficients stay
  within 1e-9 of the oracle's (the contract is 1e-6), the status array is the reference's."""
  import torch
  from oracle import oracle
  F, L, order = 16384 + 64, 480, 16
  sig = np.random.default_rng(77).uniform(-1, 1, F * L)
  sig[5 * L:6 * L] = 0.0
  d = torch.from_numpy(sig).cuda()
  c, e, st = lpc.kautocor_frames(d, L, order, fused=True)
  c0, e0, st0 = lpc.kautocor_frames(d, L, order)
  rc, re, rs = oracle.kautocor_frames(sig, F, L, L, order)
  st = st.cpu().numpy()
  assert np.array_equal(st, rs) and np.array_equal(st0.cpu().numpy(), rs)
  ok = st == 0
  c, e = c.cpu().numpy(), e.cpu().numpy()
  scale = np.abs(rc[ok]).max(axis=1, keepdims=True)
  assert (np.abs(c[ok] - rc[ok]) / scale).max() <= TOL
  assert (np.abs(e[ok] - re[ok]) / np.abs(re[ok])).max() <= TOL
  assert not np.array_equal(c[ok], c0.cpu().numpy()[ok])       # it really is the other arithmetic


# ---- ALZ_LPC_DENSE: the reference's dense Levinson-Durbin, bit-identical ---------------------------
def same_bits(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return a.shape == b.shape and bool(np.array_equal(a.view(np.uint64), b.view(np.uint64)))


def test_golden_kautocor_exact_mode_is_bit_identical_to_the_reference(lpc):
  """exact=True: coefficients and .error equal the reference's own outputs (tests/golden/lpc.json,
  generated by running the reference) in every bit."""
  for case in load_golden("lpc.json")["kautocor"]:
    x = np.array(unhex(case["x"]))
    ref = np.array(unhex(case["coefs"]))
    c, e, st = lpc.kautocor_frames(x, len(x), case["order"], exact=True)
    ref = np.concatenate([ref, np.zeros(c.shape[1] - len(ref))])
    assert st[0] == 0
    assert same_bits(c[0], ref), (case["order"], c[0], ref)
    assert same_bits([e[0]], [unhex(case["error"])])


def test_golden_levinson_exact(lpc):
  for case in load_golden("lpc.json").get("levinson", []):
    ac = unhex(case["ac"])
    filt = lpc.levinson_durbin(ac, case.get("order"))
    ref = unhex(case["coefs"])
    got = filt.numlist
    assert same_bits(got + [0.0] * (len(ref) - len(got)), ref + [0.0] * (len(got) - len(ref)))
    assert same_bits([filt.error], [unhex(case["error"])])


@pytest.mark.parametrize("order", [1, 2, 3, 4, 6, 8, 10, 12, 16, 5, 20, 40])
def test_exact_mode_vs_oracle_all_frames(lpc, order):
  """Unrolled kernels (curated orders) and the run-time-loop kernel (5, 20, 40) against the C oracle on
  every frame, ParCorError frames included (status only)."""
  from oracle import oracle
  from audiolazy_amd import _ffi
  rng = np.random.default_rng(order)
  F, L = 2048 + 3, 96
  sig = rng.uniform(-1, 1, F * L)
  sig[5 * L:6 * L] = 0.0                                   # a silent frame
  sig[9 * L:10 * L] = np.tile([1.0, -1.0], L // 2)         # lag pattern with exact cancellations
  c, e, st = lpc.kautocor_frames(sig, L, order, exact=True)
  rc, re, rs = oracle.kautocor_frames(sig, F, L, L, order)
  assert np.array_equal(st, rs) and st[5] == _ffi.E_PARCOR
  ok = st == 0
  assert same_bits(c[ok], rc[ok])
  assert same_bits(e[ok], re[ok])


@pytest.mark.parametrize("order,waves", [(16, 5), (8, 5), (12, 5), (16, 256), (8, 256)])
def test_exact_mode_waves_that_leave_the_regular_step(lpc, order, waves):
  """The dense Levinson-Durbin's select-free step (round 6) is taken while EVERY lane of a wave is regular; these frames make single
  lanes irregular at different steps, in different waves: an impulse (every reflection coefficient exactly zero: the dense
  coefficient list shrinks from step 1 on), two pulses three samples apart (zeros at steps 1 - 2, not at 3), a silent frame
  (ParCorError at step 1), a frame that is silent but for its last sample, and a constant frame -- each among random frames,
  with a ragged last wave.  Coefficients, error and status of every frame against the oracle, bit for bit.  5 waves: lags and
  Levinson-Durbin in two launches (k_levinson_dense, partial last wave); 256 waves: the one-launch staged kernel."""
  from oracle import oracle
  from audiolazy_amd import _ffi
  rng = np.random.default_rng(100 + order)
  F, L = 64 * waves + 7, 96
  sig = rng.uniform(-1, 1, F * L)
  def frame(i):
    return sig[i * L:(i + 1) * L]
  frame(3)[:] = 0.0; frame(3)[0] = 1.0                      # impulse, wave 0
  frame(70)[:] = 0.0; frame(70)[10] = 1.0; frame(70)[13] = 0.5   # wave 1: lags 1, 2 exactly zero
  frame(131)[:] = 0.0                                        # wave 2: silent
  frame(200)[:] = 0.0; frame(200)[L - 1] = -0.75             # wave 3
  frame(F - 5)[:] = 0.25                                     # ragged last wave: constant
  c, e, st = lpc.kautocor_frames(sig, L, order, exact=True)
  rc, re, rs = oracle.kautocor_frames(sig, F, L, L, order)
  assert np.array_equal(st, rs) and st[131] == _ffi.E_PARCOR
  ok = st == 0
  assert ok[3] and ok[70] and ok[200]
  assert same_bits(c[ok], rc[ok])
  assert same_bits(e[ok], re[ok])
  assert same_bits(c[3], np.r_[1.0, np.zeros(order)])        # the impulse's filter is 1


def test_exact_mode_full_width_cfg5(lpc):
  """configs[4] at full width: 65 536 frames x 480 samples, order 16, bit-identical on every frame."""
  import torch
  from oracle import oracle
  F, L, order = 65536, 480, 16
  sig = np.random.default_rng(9).uniform(-1, 1, F * L)
  sig[7 * L:8 * L] = 0.0
  d = torch.from_numpy(sig).cuda()
  c, e, st = lpc.kautocor_frames(d, L, order, exact=True)
  rc, re, rs = oracle.kautocor_frames(sig, F, L, L, order)
  st = st.cpu().numpy()
  assert np.array_equal(st, rs) and np.count_nonzero(st) == 1
  ok = st == 0
  assert same_bits(c.cpu().numpy()[ok], rc[ok])
  assert same_bits(e.cpu().numpy()[ok], re[ok])


def test_levinson_zero_extended_lags_exact(lpc):
  """order >= len(acdata): the lag list is zero-extended (lazy_lpc.py:117-118); exact zeros at the top of
  the coefficient list shrink the dense list like Poly does."""
  from oracle import oracle
  ac = [3.0, 1.0, 0.5]
  filt = lpc.levinson_durbin(ac, 5)
  rc, re = oracle.levinson_durbin(ac, 5)
  got = filt.numlist + [0.0] * (6 - len(filt.numlist))
  assert same_bits(got, rc)
  assert same_bits([filt.error], [re])


def test_preallocated_results(lpc):
  import torch
  F, L, order = 300, 64, 8
  sig = torch.rand(F * L, dtype=torch.float64, device="cuda") * 2 - 1
  c0, e0, s0 = lpc.kautocor_frames(sig, L, order)
  out = (torch.empty((F, order + 1), dtype=torch.float64, device="cuda"), torch.empty((F,), dtype=torch.float64, device="cuda"),
         torch.empty((F,), dtype=torch.int32, device="cuda"))
  c1, e1, s1 = lpc.kautocor_frames(sig, L, order, out=out)
  assert c1 is out[0] and e1 is out[1] and s1 is out[2]
  assert torch.equal(c0, c1) and torch.equal(e0, e1) and torch.equal(s0, s1)
  with pytest.raises(ValueError):
    lpc.kautocor_frames(sig, L, order, out=(out[0][:10], out[1], out[2]))


def test_lpc_strategies_match_the_reference(lpc):
  """``lpc`` (= ``lpc.autocor``), ``lpc.nautocor``, ``lpc.kautocor`` past order 63 and ``acorr`` past 63 lags against
  tests/golden/lpc_strategies.json, generated by the reference (round-2 advisor: none of this was covered, and
  ``lpc(blk, 100)`` raised NotImplementedError)."""
  import audiolazy_amd as alz
  g = load_golden("lpc_strategies.json")
  blocks = {k: unhex(v) for k, v in g["blocks"].items()}
  pad = lambda got, n: list(got) + [0.] * (n - len(got))
  bits = lambda a: np.asarray(a, dtype=np.float64).view(np.uint64)
  for case in g["acorr"]:
    assert np.array_equal(bits(alz.acorr(blocks[case["blk"]], case["max_lag"])), bits(unhex(case["r"])))
  for case in g["kautocor"]:
    f = alz.lpc.kautocor(blocks[case["blk"]], case["order"])
    ref = unhex(case["coefs"])
    assert np.array_equal(bits(pad(f.numlist, len(ref))), bits(ref)) and f.error == unhex(case["error"])
  for case in g["nautocor"]:
    f = alz.lpc.nautocor(blocks[case["blk"]], case["order"])
    ref = unhex(case["coefs"])
    np.testing.assert_allclose(pad(f.numlist, len(ref)), ref, rtol=1e-9, atol=1e-9)
    assert f.error == pytest.approx(unhex(case["error"]), rel=1e-9, abs=1e-9)
  for case in g["autocor"]:
    f = alz.lpc(blocks[case["blk"]], case["order"])
    ref = unhex(case["coefs"])
    if case["route"] == "kautocor":        # order >= 100: Levinson-Durbin, bit-identical
      assert np.array_equal(bits(pad(f.numlist, len(ref))), bits(ref)) and f.error == unhex(case["error"])
    else:                                  # pseudo-inverse form (also the ParCorError fallback of the silent block)
      np.testing.assert_allclose(pad(f.numlist, len(ref)), ref, rtol=1e-9, atol=1e-9)
      assert f.error == pytest.approx(unhex(case["error"]), rel=1e-9, abs=1e-9)
  # order defaults to len(blk) - 1 (lazy_lpc.py:154-155)
  short = blocks["noise"][:12]
  assert alz.lpc(short).numlist == alz.lpc(short, 11).numlist
  assert alz.lpc.kautocor(short).numlist == alz.lpc.kautocor(short, 11).numlist
  with pytest.raises(alz.ParCorError):
    alz.lpc.kautocor(blocks["silent"], 100)


def test_bit_identical_batch_is_one_launch(lpc):
  """exact=True on a large batch: the dense Levinson-Durbin runs inside k_acorr_stage (one launch) and is still
  bit-identical to the oracle on every frame, the silent one included; batches too small for the staged kernel
  and orders without an unrolled kernel take the two-launch form with the same doubles."""
  from oracle import oracle
  rng = np.random.default_rng(77)
  for F, L, order in ((16384 + 37, 480, 16), (16384, 200, 12), (300, 480, 16), (16400, 480, 14)):
    sig = rng.uniform(-1, 1, F * L)
    sig[3 * L: 4 * L] = 0.0
    c, e, st = lpc.kautocor_frames(sig, L, order, exact=True)
    rc, re, rs = oracle.kautocor_frames(sig, F, L, L, order)
    assert np.array_equal(st, rs) and st[3] == -4
    ok = st == 0                                              # (a ParCorError frame has a status, not coefficients)
    assert np.array_equal(c[ok].view(np.uint64), rc[ok].view(np.uint64))
    assert np.array_equal(e[ok].view(np.uint64), re[ok].view(np.uint64))
