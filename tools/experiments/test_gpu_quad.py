"""k_quad (audiolazy_amd/csrc/alz_quad.inc): a four-section cascade inside one wave -- sections in the DPP banks,
hand-over by register moves.  Every case bit for bit against the C oracle (the reference's generated DF-I statement,
lazy_filters.py:197-260, section after section, lazy_filters.py:988-990)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def al():
  import audiolazy_amd
  assert audiolazy_amd.device_count() >= 1
  return audiolazy_amd


@pytest.fixture(scope="module")
def oracle():
  from oracle import oracle as o
  return o


def same_bits(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return a.shape == b.shape and bool(np.all(a.view(np.uint64) == b.view(np.uint64)))


def _sections(rng, C, taps):
  """Four sections with the numerator taps `taps` (delays present) and both feedback taps, per-channel sets."""
  nb = max(taps) + 1
  secs = []
  for _ in range(4):
    b = np.zeros((C, nb))
    for k in taps:
      b[:, k] = rng.uniform(.2, 1., C) * rng.choice([-1., 1.], C)
    r, w = rng.uniform(.5, .97, C), rng.uniform(.05, 3., C)
    a = np.stack([np.ones(C), -2 * r * np.cos(w), r * r], axis=1)
    secs.append((b, a))
  return secs, nb


@pytest.mark.parametrize("layout", ["time", "chan"])
@pytest.mark.parametrize("taps", [(0,), (0, 1), (0, 2), (0, 1, 2)])
@pytest.mark.parametrize("N", [16 + 3, 35, 64, 80, 16 * 7 + 1, 16 * 12, 16 * 33 + 9])
def test_quad_patterns_lengths_layouts(al, oracle, layout, taps, N):
  """Tile counts 1 .. 33 (no fast tile, fast tiles with and without left-over tiles, a ragged tail), the four tap
  patterns, both layouts, non-zero initial state, and the next block continuing from the state k_quad left."""
  rng = np.random.default_rng(100 * N + 10 * len(taps) + taps[-1] + (layout == "time"))
  C = 80
  secs, nb = _sections(rng, C, taps)
  bank = al.FilterBank(secs, n_inputs=C)
  bank.reset(memory=[0.25, -0.5], zero=0.125)
  x = rng.uniform(-1, 1, (N, C))
  x2 = rng.uniform(-1, 1, (37, C))
  y = bank.process(x if layout == "time" else np.ascontiguousarray(x.T), layout=layout)
  assert "k_quad" in bank.last_kernel, bank.last_kernel
  y2 = bank.process(x2 if layout == "time" else np.ascontiguousarray(x2.T), layout=layout)
  if layout == "chan":
    y, y2 = y.T, y2.T
  bcat = np.concatenate([s[0] for s in secs], axis=1)
  acat = np.concatenate([s[1] for s in secs], axis=1)
  xh = np.full((C, max(4 * (nb - 1), 1)), 0.125)
  yh = np.tile(np.array([0.25, -0.5] * 4), (C, 1))
  ref = oracle.bank([nb] * 4, [3] * 4, bcat, acat, np.concatenate([x, x2]), xh=xh, yh=yh, zero=0.125)
  assert same_bits(y, ref[:N]), (layout, taps, N)
  assert same_bits(y2, ref[N:]), (layout, taps, N, "continuation")


@pytest.mark.parametrize("layout", ["time", "chan"])
def test_quad_outer_bank_by_input(al, oracle, layout):
  """An OUTER bank read by input index (channel = band * n_inputs + stream), 3 bands x 128 streams, shared sets per band."""
  rng = np.random.default_rng(77)
  B, S, N = 3, 128, 16 * 21 + 4
  secs, nb = _sections(rng, B, (0, 1))
  bank = al.FilterBank(secs, n_inputs=S, mode="outer")
  bank.reset()
  x = rng.uniform(-1, 1, (S, N))
  y = bank.process(x if layout == "chan" else np.ascontiguousarray(x.T), layout=layout)
  assert "k_quad" in bank.last_kernel, bank.last_kernel
  if layout == "time":
    y = y.T
  bcat = np.repeat(np.concatenate([s[0] for s in secs], axis=1), S, axis=0)
  acat = np.repeat(np.concatenate([s[1] for s in secs], axis=1), S, axis=0)
  ref = oracle.bank([nb] * 4, [3] * 4, bcat, acat, np.tile(x, (B, 1)), layout="chan")
  assert same_bits(y, ref)


def test_quad_fma_mode_is_the_other_arithmetic(al, oracle):
  rng = np.random.default_rng(5)
  C, N = 128, 16 * 40
  secs, nb = _sections(rng, C, (0, 1))
  x = rng.uniform(-1, 1, (N, C))
  exact = al.FilterBank(secs, n_inputs=C)
  exact.reset()
  y = exact.process(x)
  fused = al.FilterBank(secs, n_inputs=C).set_fused(True)
  fused.reset()
  yf = fused.process(x)
  assert "k_quad" in fused.last_kernel and "fma" in fused.last_kernel, fused.last_kernel
  scale = np.maximum(np.abs(y).max(axis=0), 1e-300)
  assert (np.abs(yf - y) / scale).max() <= 1e-10
  assert not same_bits(yf, y)
