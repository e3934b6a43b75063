"""k_mid (csrc/alz_mid.hip, round 6): one-section IIR filters of order 3 .. 8 with dense coefficients -- ZFilter(butter(...)) as
the reference's examples/butterworth_with_noise.py:52-67 builds them --, all-pole filters, longer numerators in front of one or
two poles, and maverage.recursive(size) (lazy_analysis.py:569-591: two numerator taps `size` apart in front of one pole) as a
three-wave streaming kernel instead of the lane-per-channel loops (k_masked / k_generic).  Bit-exact against the oracle as ONE
continuous run over blocks of whole tiles, ragged tiles and less than a tile, from a non-trivial state, asserting the kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def alz():
  import audiolazy_amd
  audiolazy_amd.load_library()
  return audiolazy_amd


@pytest.fixture(scope="module")
def oracle():
  from oracle import oracle
  return oracle


def same_bits(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return a.shape == b.shape and bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


def stable_den(rng, C, K):
  """[C, K + 1] denominators with a0 = 1 and all roots inside the unit circle (random pole pairs, one real pole when K is odd)."""
  out = np.zeros((C, K + 1))
  for c in range(C):
    poles = []
    for _ in range(K // 2):
      r, w = rng.uniform(0.5, 0.97), rng.uniform(0.05, 3.0)
      poles += [r * np.exp(1j * w), r * np.exp(-1j * w)]
    if K % 2:
      poles.append(rng.uniform(-0.9, 0.95))
    out[c] = np.real(np.poly(poles))
    out[c, 0] = 1.0
  return out


def run_blocks(alz, oracle, b, a, lens, C, expect, inplace=False, n_inputs=None, mode="diagonal", rng=None):
  import torch
  rng = rng or np.random.default_rng(1)
  nb, na = b.shape[-1], a.shape[-1]
  n_in = n_inputs or C
  xs = [rng.uniform(-1, 1, (m, n_in)) for m in lens]
  xh0 = rng.uniform(-1, 1, (C, max(nb - 1, 1)))
  yh0 = rng.uniform(-1, 1, (C, max(na - 1, 1)))
  bank = alz.FilterBank([(b, a)], n_inputs=n_in, mode=mode)
  bank.set_state(xh0, yh0)
  xall = np.concatenate(xs)
  if mode == "outer":
    sets = b.shape[0]
    ref = np.concatenate([oracle.bank([nb], [na], b[s], a[s], xall, layout="time", xh=xh0[s * n_in:(s + 1) * n_in].copy(),
                                      yh=yh0[s * n_in:(s + 1) * n_in].copy()) for s in range(sets)], axis=1)
  else:
    ref = oracle.bank([nb], [na], b, a, xall, layout="time", xh=xh0.copy(), yh=yh0.copy())
  at = 0
  for x in xs:
    xd = torch.from_numpy(x).cuda()
    y = bank.process(xd, layout="time", out=xd if inplace else None)
    if x.shape[0] >= 64:
      assert expect in bank.last_kernel, (bank.last_kernel, expect)
    assert same_bits(y.cpu().numpy(), ref[at:at + x.shape[0]]), (expect, at, bank.last_kernel)
    at += x.shape[0]


LENS = [64 * 5 + 13, 64, 30, 64 * 3, 7, 64 * 9 + 63]


@pytest.mark.parametrize("order", [3, 4, 5, 6, 7, 8])
def test_dense_iir_sections_of_order_3_to_8(alz, oracle, order):
  rng = np.random.default_rng(order)
  C = 48
  a = stable_den(rng, C, order)
  b = rng.uniform(-1, 1, (C, order + 1))
  run_blocks(alz, oracle, b, a, LENS, C, "k_mid", rng=rng)
  run_blocks(alz, oracle, b, a, LENS, C, "k_mid", inplace=True, rng=rng)
  # one coefficient set shared by the bank
  run_blocks(alz, oracle, b[0], a[0], LENS[:3], C, "k_mid", rng=rng)


def test_butterworth_sections_as_the_reference_example_builds_them(alz, oracle):
  """examples/butterworth_with_noise.py:52-67: ZFilter(butter(order, cutoff)) for order 4 and 6, here a bank of cutoffs."""
  from scipy import signal
  rng = np.random.default_rng(46)
  C = 64
  for order in (4, 6):
    ba = [signal.butter(order, wn) for wn in np.linspace(0.08, 0.6, C)]
    b, a = np.array([x[0] for x in ba]), np.array([x[1] for x in ba])
    assert np.all(a[:, 0] == 1.0)
    run_blocks(alz, oracle, b, a, [64 * 40 + 5, 64 * 3], C, "k_mid", rng=rng)


@pytest.mark.parametrize("K", [3, 4, 6, 8])
def test_all_pole_sections(alz, oracle, K):
  rng = np.random.default_rng(10 + K)
  C = 32
  run_blocks(alz, oracle, rng.uniform(0.1, 1, (C, 1)), stable_den(rng, C, K), LENS, C, "k_mid", rng=rng)


@pytest.mark.parametrize("nb,K", [(4, 1), (6, 1), (9, 1), (4, 2), (7, 2), (9, 2)])
def test_longer_numerators_in_front_of_one_or_two_poles(alz, oracle, nb, K):
  rng = np.random.default_rng(100 * nb + K)
  C = 32
  run_blocks(alz, oracle, rng.uniform(-1, 1, (C, nb)), stable_den(rng, C, K), LENS, C, "k_mid", rng=rng)


@pytest.mark.parametrize("size", [9, 40, 64, 100, 255, 256])
def test_maverage_recursive_far_tap(alz, oracle, size):
  """maverage.recursive(size) (lazy_analysis.py:569-591): y[n] = y[n-1] + (x[n] - x[n-size]) / size as the reference's
  z-algebra writes it -- b0 = 1/size, b_size = -1/size, a1 = -1."""
  rng = np.random.default_rng(size)
  C = 48
  b = np.zeros(size + 1)
  b[0], b[size] = 1.0 / size, -1.0 / size
  a = np.array([1.0, -1.0])
  run_blocks(alz, oracle, b, a, [64 * 7 + 20, 64, 64 * 2 + 1, 10, 64 * 12], C, "k_mid<far tap>", rng=rng)
  run_blocks(alz, oracle, b, a, [64 * 7 + 20, 64], C, "k_mid<far tap>", inplace=True, rng=rng)
  # per-channel gains and a leaky pole, a second dense tap
  bb = np.zeros((C, size + 1))
  bb[:, 0], bb[:, 1], bb[:, size] = rng.uniform(0.5, 1, C), rng.uniform(-0.2, 0.2, C), rng.uniform(-1, -0.5, C)
  aa = np.stack([np.ones(C), -rng.uniform(0.9, 0.999, C), rng.uniform(0.0, 0.05, C)], axis=1)
  run_blocks(alz, oracle, bb, aa, [64 * 5 + 3, 64 * 6], C, "k_mid<far tap>", rng=rng)


def test_outer_bank_and_wide_bank(alz, oracle):
  rng = np.random.default_rng(77)
  S, B, order = 16, 3, 4
  a = stable_den(rng, B, order)
  b = rng.uniform(-1, 1, (B, order + 1))
  run_blocks(alz, oracle, b, a, [64 * 4 + 9, 64 * 2], B * S, "k_mid", n_inputs=S, mode="outer", rng=rng)
  # a bank as wide as bench.py's (4096 channels), short
  C = 4096
  run_blocks(alz, oracle, rng.uniform(-1, 1, (C, 7)), stable_den(rng, C, 6), [64 * 6], C, "k_mid", rng=rng)


def test_band_pass_butterworth_zero_taps_inside_the_numerator(alz, oracle):
  """scipy.signal.butter(n, [lo, hi], "bandpass"): b = [b0, 0, -n b0, 0, ...] -- zero taps inside a short numerator are absent
  from the sum (lazy_filters.py:205-206), the denominator is dense: k_mid with the numerator's pattern as a scalar mask."""
  from scipy import signal
  rng = np.random.default_rng(12)
  C = 32
  for n in (2, 3, 4):
    ba = [signal.butter(n, [lo, lo + 0.2], "bandpass") for lo in np.linspace(0.1, 0.5, C)]
    b, a = np.array([x[0] for x in ba]), np.array([x[1] for x in ba])
    b[np.abs(b) < 1e-300] = 0.0
    assert np.all(b[:, 1] == 0) and np.all(a[:, 0] == 1.0) and np.all(a[:, 1:] != 0)
    run_blocks(alz, oracle, b, a, [64 * 6 + 9, 64 * 2, 30], C, "k_mid", rng=rng)
  # a leading zero tap (a pure delay in front) and a random pattern
  b = rng.uniform(-1, 1, (C, 6)); b[:, 0] = 0.0; b[:, 3] = 0.0
  run_blocks(alz, oracle, b, stable_den(rng, C, 5), [64 * 5 + 1, 64], C, "k_mid", rng=rng)


def test_shapes_outside_stay_on_the_other_kernels(alz, oracle):
  """a0 != 1, a zero inside the denominator, a ragged channel count: the lane-per-channel kernels as before, same doubles."""
  rng = np.random.default_rng(5)
  C = 32
  a = stable_den(rng, C, 4)
  b = rng.uniform(-1, 1, (C, 5))
  a2 = a.copy(); a2[:, 0] = 2.0
  run_blocks(alz, oracle, b, a2, [64 * 3], C, "k_masked", rng=rng)
  a3 = a.copy(); a3[:, 2] = 0.0                       # a zero inside the DENOMINATOR: the recurrence wave's chain is dense only
  run_blocks(alz, oracle, b, a3, [64 * 3], C, "k_masked", rng=rng)
  run_blocks(alz, oracle, b[:20], a[:20], [64 * 3], 20, "k_masked", rng=rng)
