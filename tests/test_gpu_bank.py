"""GPU parity: the HIP filter bank (through the C ABI) against the committed
golden vectors of the reference and against the CPU oracle on seeded inputs.

Bar: BIT-EXACT.  The path is float64 and the north star allows 1e-6 relative,
but the DF-I kernels follow the reference's generated expression term by term
with unfused multiply/add, so every comparison here is on the raw 64-bit
patterns (tolerance 0), signed zeros included.
"""
import numpy as np
import pytest

from conftest import load_golden, unhex

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def alz():
  import audiolazy_amd
  audiolazy_amd.load_library()
  assert audiolazy_amd.device_count() >= 1, "no HIP device: the engine has no CPU path"
  return audiolazy_amd


@pytest.fixture(scope="module")
def oracle():
  from oracle import oracle as o
  return o


def same_bits(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return a.shape == b.shape and bool(
    np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


FILT = load_golden("filters.json")


@pytest.mark.parametrize("case", FILT["cases"], ids=lambda c: c["tag"])
def test_golden_single_filters(alz, case):
  base = unhex(FILT["x"])
  x = unhex(case["x"]) if "x" in case else base[:case["x_len"]]
  mem = None if case["memory"] is None else unhex(case["memory"])
  bank = alz.FilterBank([(unhex(case["b"]), unhex(case["a"]))], n_inputs=1)
  y = list(bank(x, memory=mem, zero=unhex(case["zero"])))
  assert same_bits(y, unhex(case["y"])), bank.last_kernel


def test_golden_lfilter_grid(alz):
  # reference tests/test_filters_extdep.py:41-47
  for case in load_golden("lfilter_grid.json"):
    bank = alz.FilterBank([(unhex(case["b"]), unhex(case["a"]))], n_inputs=1)
    assert same_bits(list(bank(unhex(case["x"]))), unhex(case["y"]))


def test_golden_cascades(alz):
  for case in load_golden("containers.json"):
    if case["kind"] != "cascade":
      continue
    secs = [(unhex(s["b"]), unhex(s["a"])) for s in case["sections"]]
    bank = alz.FilterBank(secs, n_inputs=1)
    mem = None if case["memory"] is None else unhex(case["memory"])
    y = list(bank(unhex(case["x"]), memory=mem, zero=unhex(case["zero"])))
    assert same_bits(y, unhex(case["y"]))


def test_golden_gammatone_cascades(alz):
  aud = load_golden("auditory.json")
  x = unhex(aud["x"])
  for g in aud["gammatone"]:
    secs = [(unhex(s["b"]), unhex(s["a"])) for s in g["sections"]]
    bank = alz.FilterBank(secs, n_inputs=1)
    assert same_bits(list(bank(x)), unhex(g["y"])), (g["strategy"], bank.last_kernel)


def test_golden_multichannel_both_layouts(alz):
  mc = load_golden("multichannel.json")
  b, a = np.array(unhex(mc["b"])), np.array(unhex(mc["a"]))
  x, y = np.array(unhex(mc["x"])), np.array(unhex(mc["y"]))
  bank = alz.FilterBank([(b, a)], n_inputs=mc["C"])
  bank.reset()
  assert same_bits(bank.process(x, layout="time"), y)
  bank.reset()
  assert same_bits(bank.process(np.ascontiguousarray(x.T), layout="chan"), y.T)
  # the reference's idiom itself: a Stream of ndarray rows
  rows = list(bank(iter(x), zero=np.zeros(mc["C"])))
  assert same_bits(np.array(rows), y)


def resonator_bank(C, rng=None):
  w = 2 * np.pi * np.geomspace(50., 20000., C) / 48000.
  r = np.exp(-w / 20.)
  a = np.stack([np.ones(C), -2 * r * np.cos(w), r * r], axis=1)
  g = (1 - r * r) / 2
  b = np.stack([g, np.zeros(C), -g], axis=1)
  return b, a


@pytest.mark.parametrize("layout", ["time", "chan"])
@pytest.mark.parametrize("C,N", [(1, 1), (3, 7), (64, 1000), (65, 257), (1000, 4099)])
def test_bank_vs_oracle_ragged_sizes(alz, oracle, layout, C, N):
  rng = np.random.default_rng(C * 7919 + N)
  b, a = resonator_bank(C)
  x = rng.uniform(-1, 1, (N, C) if layout == "time" else (C, N))
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  bank.reset()
  y = bank.process(x, layout=layout)
  assert any(k in bank.last_kernel for k in ("k_small", "k_wave", "k_duo"))
  assert same_bits(y, oracle.bank([3], [3], b, a, x, layout=layout))


def test_empty_block(alz):
  b, a = resonator_bank(4)
  bank = alz.FilterBank([(b, a)], n_inputs=4)
  bank.reset()
  assert bank.process(np.zeros((0, 4))).shape == (0, 4)
  assert list(bank([])) == []


def test_torch_device_path_and_block_continuity(alz, oracle):
  import torch
  rng = np.random.default_rng(11)
  C, N = 512, 6000
  b, a = resonator_bank(C)
  x = rng.uniform(-1, 1, (N, C))
  ref = oracle.bank([3], [3], b, a, x)
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  bank.reset()
  xd = torch.from_numpy(x).cuda()
  y = bank.process(xd).cpu().numpy()
  assert same_bits(y, ref)
  # ragged consecutive blocks == one continuous run (state carried on the device)
  bank.reset()
  parts, pos = [], 0
  for n in (1, 7, 64, 1000, 2, 4926):
    parts.append(bank.process(xd[pos:pos + n].contiguous()).cpu().numpy())
    pos += n
  assert pos == N and same_bits(np.concatenate(parts), ref)
  # state round trip: get_state after the run == last samples
  xh, yh = bank.get_state()
  assert same_bits(xh, x[-1:-3:-1].T) and same_bits(yh, ref[-1:-3:-1].T)
  bank2 = alz.FilterBank([(b, a)], n_inputs=C)
  bank2.set_state(xh, yh)
  x2 = rng.uniform(-1, 1, (50, C))
  assert same_bits(bank2.process(x2), bank.process(x2))


def test_in_place_diagonal(alz, oracle):
  import torch
  rng = np.random.default_rng(12)
  C, N = 128, 999
  b, a = resonator_bank(C)
  x = rng.uniform(-1, 1, (N, C))
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  bank.reset()
  xd = torch.from_numpy(x).cuda()
  bank.process(xd, out=xd)
  assert same_bits(xd.cpu().numpy(), oracle.bank([3], [3], b, a, x))


def test_mixed_zero_patterns_use_masked_kernel(alz, oracle):
  rng = np.random.default_rng(13)
  C, N = 96, 500
  b = rng.uniform(-1, 1, (C, 3))
  a = np.concatenate([np.ones((C, 1)), rng.uniform(-.4, .4, (C, 2))], axis=1)
  b[::3, 1] = 0.0
  b[1::4, 0] = 0.0
  a[::5, 2] = 0.0
  a[2::7, 1] = 0.0
  a[::2, 0] = rng.uniform(.5, 2., C // 2)          # per-channel gain
  b[5], a[5, 1:] = 0.0, 0.0                          # an all-zero channel -> yields `zero`
  x = rng.uniform(-1, 1, (N, C))
  x[17, 3] = np.inf                                  # zero taps must stay absent from the sum
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  bank.reset(zero=0.25)
  y = bank.process(x)
  assert bank.last_kernel == "k_masked<3,3>"
  ref = oracle.bank([3], [3], b, a, x, zero=0.25,
                    xh=np.full((C, 2), .25), yh=np.full((C, 2), .25))
  assert same_bits(y, ref)
  assert np.all(y[:, 5] == 0.25)


@pytest.mark.parametrize("nb,na,kern", [(8, 3, "k_masked<8,3>"), (12, 7, "k_masked<16,9>"),
                                        (40, 12, "k_generic"), (300, 1, "k_fir")])
def test_higher_orders(alz, oracle, nb, na, kern):
  rng = np.random.default_rng(nb * 100 + na)
  C, N = 70, 700
  b = rng.uniform(-1, 1, (C, nb))
  a = np.concatenate([np.ones((C, 1)), rng.uniform(-1, 1, (C, na - 1)) * (0.5 / max(na - 1, 1))], axis=1)
  x = rng.uniform(-1, 1, (N, C))
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  bank.reset(memory=[0.5, -0.25, 0.125], zero=0.0625)
  y = bank.process(x)
  assert kern in bank.last_kernel
  hist = [alz.memory_to_hist([0.5, -0.25, 0.125], na - 1, 0.0625)]
  yh = np.repeat(np.array(hist), C, axis=0).reshape(C, -1) if na > 1 else np.zeros((C, 1))
  ref = oracle.bank([nb], [na], b, a, x, xh=np.full((C, max(nb - 1, 1)), 0.0625),
                    yh=np.ascontiguousarray(yh, dtype=float), zero=0.0625)
  assert same_bits(y, ref)
  # second block continues the stream
  x2 = rng.uniform(-1, 1, (33, C))
  whole = oracle.bank([nb], [na], b, a, np.concatenate([x, x2]),
                      xh=np.full((C, max(nb - 1, 1)), 0.0625),
                      yh=np.ascontiguousarray(yh, dtype=float), zero=0.0625)
  assert same_bits(bank.process(x2), whole[N:])


def test_shared_coefficients_and_outer_mode(alz, oracle):
  rng = np.random.default_rng(21)
  S, B, N = 24, 5, 900
  x = rng.uniform(-1, 1, (N, S))
  # shared: one set on every input
  b1, a1 = [0.2, 0.0, -0.2], [1.0, -1.6, 0.81]
  bank = alz.FilterBank([(b1, a1)], n_inputs=S)
  bank.reset()
  assert same_bits(bank.process(x), oracle.bank([3], [3], np.array(b1), np.array(a1), x))
  # outer: B cascades x S inputs -> [N, B*S], channel = band*S + stream
  bb, aa = resonator_bank(B)
  secs = [(bb, aa), (bb[:, :1] * 3.0, aa)]
  fb = alz.FilterBank(secs, n_inputs=S, mode="outer")
  fb.reset()
  y = fb.process(x)
  assert y.shape == (N, B * S)
  for band in range(B):
    ref = oracle.bank([3, 1], [3, 3], np.concatenate([bb[band], bb[band, :1] * 3.0]),
                      np.concatenate([aa[band], aa[band]]), x)
    assert same_bits(y[:, band * S:(band + 1) * S], ref)
  # channel-major: x [S, N] -> y [B*S, N]  ( == [B, S, N] )
  fb.reset()
  yc = fb.process(np.ascontiguousarray(x.T), layout="chan")
  assert same_bits(yc, y.T)


def test_errors_match_the_reference(alz):
  with pytest.raises(ZeroDivisionError):          # lazy_filters.py:177-178
    alz.FilterBank([([1.], [0., 1.])], n_inputs=1)
  bank = alz.FilterBank([([1.], [1., -.5])], n_inputs=2)
  with pytest.raises(ValueError):
    bank.process(np.zeros((4, 3)))


def test_full_scale_properties(alz):
  """BASELINE cfg2 width (4096 channels), long block: size-independent checks --
  linearity in the input and block-split invariance on the device."""
  import torch
  C, N = 4096, 1 << 14
  b, a = resonator_bank(C)
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  g = torch.Generator(device="cuda").manual_seed(5)
  x = torch.rand((N, C), dtype=torch.float64, device="cuda", generator=g) * 2 - 1
  bank.reset()
  y = bank.process(x)
  bank.reset()
  y2 = torch.cat([bank.process(x[:5000].contiguous()), bank.process(x[5000:].contiguous())])
  assert torch.equal(y, y2)
  bank.reset()
  y4 = bank.process(x * 4.0)            # scaling by a power of two is exact in binary64
  assert torch.equal(y4, y * 4.0)
  assert torch.isfinite(y).all()


def hamming_sinc(ntaps=256, fc=0.1):
  """256-point Hamming-windowed sinc lowpass at 0.1 fs, plain float64."""
  import math
  m = (ntaps - 1) / 2.0
  taps = []
  for i in range(ntaps):
    t = i - m
    ideal = 2 * fc if t == 0 else math.sin(2 * math.pi * fc * t) / (math.pi * t)
    taps.append(ideal * (0.54 - 0.46 * math.cos(2 * math.pi * i / (ntaps - 1))))
  return taps


@pytest.mark.parametrize("C,N", [(64, 300), (200, 1000), (130, 77)])
def test_fir256_shared_taps_bit_exact(alz, oracle, C, N):
  rng = np.random.default_rng(C + N)
  taps = np.array(hamming_sinc())
  x = rng.uniform(-1, 1, (N, C))
  bank = alz.FilterBank([(taps, [1.0])], n_inputs=C)
  bank.reset(zero=0.125)
  y = bank.process(x)
  assert bank.last_kernel == "k_fir_ring"
  ref = oracle.bank([256], [1], taps, np.array([1.0]), x, xh=np.full((C, 255), 0.125), zero=0.125)
  assert same_bits(y, ref)
  # blocks continue the stream (history kept on the device), including blocks shorter than the taps
  x2 = rng.uniform(-1, 1, (40, C))
  x3 = rng.uniform(-1, 1, (500, C))
  whole = oracle.bank([256], [1], taps, np.array([1.0]), np.concatenate([x, x2, x3]),
                      xh=np.full((C, 255), 0.125), zero=0.125)
  assert same_bits(bank.process(x2), whole[N:N + 40])
  assert same_bits(bank.process(x3), whole[N + 40:])
  # channel-major blocks: one channel per wave, window staged in LDS -- same numbers
  bank.reset(zero=0.125)
  yc = bank.process(np.ascontiguousarray(x.T), layout="chan")
  assert bank.last_kernel == "k_fir_cm" and same_bits(yc, ref.T)
  x2c = np.ascontiguousarray(x2.T)
  assert same_bits(bank.process(x2c, layout="chan"), whole[N:N + 40].T)


@pytest.mark.parametrize("nb,N,gain", [(20, 5, 1.0), (17, 77, 1.0), (64, 300, 2.0), (255, 1000, 1.0),
                                       (256, 40, -1.0), (100, 2100, 0.5), (256, 4099, 1.0), (24, 1537, 1.0)])
def test_fir_shared_taps_kernels_shapes(alz, oracle, nb, N, gain):
  """The shared-tap ring kernel (edge runs that reach into the delay line, interior runs, the interleaved
  run-to-wave mapping) on tap counts that are not a multiple of the tap block, blocks shorter than a row
  group and blocks that end inside a run, zero taps (+0 and -0: absent from the sum, an inf must not get
  through them) and a gain.  (The round-1 variants k_fir_s / k_fir<shared> are no longer selectable at run
  time: the shipped library has no tuning switches.)"""
  name = "k_fir_ring"
  rng = np.random.default_rng(nb * 7 + N)
  C = 70
  taps = rng.uniform(-1, 1, nb)
  taps[3] = 0.0
  taps[nb - 2] = -0.0
  x = rng.uniform(-1, 1, (N, C))
  if N > 10:
    x[N // 2, 11] = np.inf                     # reaches y only through taps that are present
  bank = alz.FilterBank([(taps, [gain])], n_inputs=C)
  bank.reset(zero=0.25)
  y = bank.process(x)
  assert bank.last_kernel == name
  x2 = rng.uniform(-1, 1, (33, C))
  whole = oracle.bank([nb], [1], taps, np.array([gain]), np.concatenate([x, x2]),
                      xh=np.full((C, nb - 1), 0.25), zero=0.25)
  assert same_bits(y, whole[:N])
  assert same_bits(bank.process(x2), whole[N:])


@pytest.mark.parametrize("nb1,nb2", [(17, 21), (18, 3), (130, 27)])
def test_fir_ring_section_followed_by_another_sections_taps(alz, oracle, nb1, nb2):
  """k_fir_ring requests the taps of a block one block ahead (round 6) and reads whole blocks of four: past the end of a section
  whose tap count is not a multiple of four that is the NEXT section's taps in the coefficient slab -- non-zero numbers that must
  not enter the sum.  A feedback-free section of nb1 taps followed by a second section, bit for bit, with a continuation block."""
  rng = np.random.default_rng(nb1 * 31 + nb2)
  C, N = 70, 700
  b1, b2 = rng.uniform(-1, 1, nb1), rng.uniform(1, 2, nb2)    # (the second section's taps: all well away from zero)
  a2 = np.array([1.0]) if nb2 > 3 else np.array([1.0, -0.5, 0.25])
  x = rng.uniform(-1, 1, (N, C))
  bank = alz.FilterBank([(b1, [1.0]), (b2, a2)], n_inputs=C)
  bank.reset()
  y = bank.process(x)
  x2 = rng.uniform(-1, 1, (50, C))
  y2 = bank.process(x2)
  whole = oracle.bank([nb1, nb2], [1, len(a2)], np.concatenate([b1, b2]), np.concatenate([[1.0], a2]), np.concatenate([x, x2]))
  assert same_bits(y, whole[:N]) and same_bits(y2, whole[N:])
  # the first section alone runs on the ring kernel (what the cascade's first stage is)
  alone = alz.FilterBank([(b1, [1.0])], n_inputs=C)
  alone.reset()
  alone.process(x)
  assert alone.last_kernel == "k_fir_ring"


def test_fir_per_channel_taps_zero_taps_and_gain(alz, oracle):
  rng = np.random.default_rng(77)
  C, N, nb = 96, 400, 40
  b = rng.uniform(-1, 1, (C, nb))
  b[::3, 5] = 0.0
  b[1::5, 0] = 0.0
  b[7] = 0.0                                   # all-zero channel -> yields `zero`
  a = rng.uniform(0.5, 2.0, (C, 1))
  x = rng.uniform(-1, 1, (N, C))
  x[100, 9] = np.inf                           # b[9, 5] == 0: the inf must not reach y through that tap
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  bank.reset(zero=-0.5)
  y = bank.process(x)
  assert bank.last_kernel == "k_fir<per-channel>"
  ref = oracle.bank([nb], [1], b, a, x, xh=np.full((C, nb - 1), -0.5), zero=-0.5)
  assert same_bits(y, ref)
  assert np.all(y[:, 7] == -0.5)


def test_fir_in_a_cascade_and_outer_mode(alz, oracle):
  rng = np.random.default_rng(5)
  S, N = 40, 600
  taps = np.array(hamming_sinc(64, 0.2))
  x = rng.uniform(-1, 1, (N, S))
  secs = [([0.5, 0.25], [1.0, -0.3]), (taps, [1.0])]   # IIR then FIR: second stage runs "in place"
  bank = alz.FilterBank(secs, n_inputs=S)
  bank.reset()
  y = bank.process(x)
  ref = oracle.bank([2, 64], [2, 1], np.concatenate([[0.5, 0.25], taps]), np.array([1.0, -0.3, 1.0]), x)
  assert same_bits(y, ref)
  two = np.stack([taps, taps[::-1] * 0.5])
  fb = alz.FilterBank([(two, [1.0])], n_inputs=S, mode="outer")
  fb.reset()
  yo = fb.process(x)
  for band in range(2):
    assert same_bits(yo[:, band * S:(band + 1) * S], oracle.bank([64], [1], two[band], np.array([1.0]), x))


def test_fir_config3_width_properties(alz):
  """BASELINE cfg3 width (8192 channels, 256 shared taps): linearity and block-split invariance."""
  import torch
  C, N = 8192, 2048
  bank = alz.FilterBank([(hamming_sinc(), [1.0])], n_inputs=C)
  g = torch.Generator(device="cuda").manual_seed(9)
  x = torch.rand((N, C), dtype=torch.float64, device="cuda", generator=g) * 2 - 1
  bank.reset()
  y = bank.process(x)
  bank.reset()
  y2 = torch.cat([bank.process(x[:700].contiguous()), bank.process(x[700:].contiguous())])
  assert torch.equal(y, y2)
  bank.reset()
  assert torch.equal(bank.process(x * 0.5), y * 0.5)
  # a unit impulse on one channel reproduces the taps (tests/test_filters.py:76-107 style)
  imp = torch.zeros((300, C), dtype=torch.float64, device="cuda")
  imp[0, 123] = 1.0
  bank.reset()
  h = bank.process(imp)[:, 123].cpu().numpy()
  assert np.array_equal(h[:256], np.array(hamming_sinc())) and np.all(h[256:] == 0)


@pytest.mark.parametrize("D,alpha", [(16, 0.5), (109, 0.97), (441, -0.9)])
def test_long_feedback_combs_use_the_step_kernel(alz, oracle, D, alpha):
  """comb.fb / comb.tau with long delays (lazy_filters.py:1090-1147) and a linearize()d-style
  two-tap feedback: steps of independent samples with the delay line as an LDS ring (round 6; round 1: the block itself
  as the delay line, lane = channel); bit-exact, state carried across blocks."""
  rng = np.random.default_rng(D)
  C, N = 80, 3 * D + 37
  x = rng.uniform(-1, 1, (N, C))
  a = np.zeros(D + 2)
  a[0], a[D], a[D + 1] = 1.0, -alpha * 0.75, -alpha * 0.25     # two adjacent feedback taps
  b = np.array([1.0, 0.0, 0.5])
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  mem = rng.uniform(-1, 1, D + 1).tolist()
  bank.reset(memory=mem, zero=0.0)
  y = bank.process(x)
  assert bank.last_kernel == "k_comb_tm"
  yh = np.tile(np.array(mem), (C, 1))
  ref = oracle.bank([3], [D + 2], b, a, x, xh=np.zeros((C, 2)), yh=yh.copy())
  assert same_bits(y, ref)
  x2 = rng.uniform(-1, 1, (D // 2 + 3, C))       # a block shorter than the delay line
  whole = oracle.bank([3], [D + 2], b, a, np.concatenate([x, x2]), xh=np.zeros((C, 2)), yh=yh.copy())
  assert same_bits(bank.process(x2), whole[N:])
  # per-channel coefficients
  al = rng.uniform(0.5, 0.99, C)
  a2 = np.zeros((C, D + 1)); a2[:, 0] = 1.0; a2[:, D] = -al
  bank2 = alz.FilterBank([(np.ones((C, 1)), a2)], n_inputs=C)
  bank2.reset()
  assert same_bits(bank2.process(x), oracle.bank([1], [D + 1], np.ones((C, 1)), a2, x)) and bank2.last_kernel == "k_comb_tm"
  # a ragged channel count stays on the lane-per-channel kernel (same doubles)
  bank3 = alz.FilterBank([(np.ones((C - 3, 1)), a2[:C - 3])], n_inputs=C - 3)
  bank3.reset()
  assert same_bits(bank3.process(x[:, :C - 3].copy()), oracle.bank([1], [D + 1], np.ones((C - 3, 1)), a2[:C - 3], x[:, :C - 3].copy()))
  assert bank3.last_kernel == "k_sparse"


def _comb_case(shape, D, C, rng):
  """(b [C, nb], a [C, na]) of the comb family: per-channel coefficients, the zero pattern of the reference's designs."""
  g = rng.uniform(0.3, 0.97, C) * rng.choice([-1.0, 1.0], C)
  f = rng.uniform(0.1, 0.9, C)
  if shape == "fb":            # comb.fb / comb.tau: 1 / (1 - g z^-D)
    b = np.ones((C, 1)); a = np.zeros((C, D + 1)); a[:, 0] = 1; a[:, D] = -g
  elif shape == "lin":         # ... .linearize(): 1 / (1 - g (1 - f) z^-D - g f z^-(D+1))   (karplus_strong)
    b = np.ones((C, 1)); a = np.zeros((C, D + 2)); a[:, 0] = 1; a[:, D] = -g * (1 - f); a[:, D + 1] = -g * f
  elif shape == "ff":          # comb.ff: 1 + g z^-D
    b = np.zeros((C, D + 1)); b[:, 0] = 1; b[:, D] = g; a = np.ones((C, 1))
  elif shape == "fflin":       # comb.ff linearized: three numerator taps
    b = np.zeros((C, D + 2)); b[:, 0] = 1; b[:, D] = g * (1 - f); b[:, D + 1] = g * f; a = np.ones((C, 1))
  else:                        # "mixed": (b0 + b3 z^-3) / (1 - g z^-D)
    b = np.zeros((C, 4)); b[:, 0] = rng.uniform(0.5, 1.5, C); b[:, 3] = rng.uniform(-0.5, 0.5, C)
    a = np.zeros((C, D + 1)); a[:, 0] = 1; a[:, D] = -g
  return b, a


@pytest.mark.parametrize("layout,C", [("time", 48), ("chan", 48), ("chan", 272)])
@pytest.mark.parametrize("shape,D", [("fb", 16), ("fb", 441), ("fb", 700), ("lin", 24), ("lin", 109), ("lin", 441), ("ff", 441),
                                       ("fflin", 70), ("mixed", 109), ("mixed", 300)])
def test_comb_step_kernels_both_layouts_in_place_and_across_blocks(alz, oracle, layout, C, shape, D):
  """Round 6: k_comb_tm / k_comb_cm (csrc/alz_comb.hip) -- the comb family of lazy_filters.py:1087-1173 and its
  linearize()d forms (:339-373) on time-major and channel-major blocks, out of place and (single numerator tap) in place,
  over blocks that are longer than, equal to and shorter than a step / a chunk / the delay line, with a non-trivial delay
  line to start from (karplus_strong's memory=white_noise, lazy_synth.py:624-657); bit-exact against the oracle as ONE
  continuous run, asserting the kernel taken."""
  import torch
  rng = np.random.default_rng(D + len(shape))
  tm = layout == "time"
  ax = 0 if tm else 1
  b, a = _comb_case(shape, D, C, rng)
  nb, na = b.shape[1], a.shape[1]
  lens = [m + (m & 1) for m in (3 * D + 38, D // 2 + 4, 2 * 256 + 78, 256, 2)]     # (even lengths: 16-byte rows in [C, N])
  xs = [rng.uniform(-1, 1, (m, C) if tm else (C, m)) for m in lens]
  xh0 = rng.uniform(-1, 1, (C, max(nb - 1, 1)))
  yh0 = rng.uniform(-1, 1, (C, max(na - 1, 1)))
  ref = oracle.bank([nb], [na], b, a, np.concatenate(xs, axis=ax), layout=layout, xh=xh0.copy(), yh=yh0.copy())
  # channel-major: up to 256 strings whose period fits a wave's registers (D <= 512, numerator b0 alone) take k_string
  want = "k_comb_tm" if tm else "k_string" if (C <= 256 and D <= 512 and shape in ("fb", "lin")) else "k_comb_cm"
  for inplace in ((False, True) if nb == 1 else (False,)):
    bank = alz.FilterBank([(b, a)], n_inputs=C)
    bank.set_state(xh0, yh0)
    at = 0
    for x in xs:
      xd = torch.from_numpy(x).cuda()
      y = bank.process(xd, layout=layout, out=xd if inplace else None)
      assert bank.last_kernel == want, (bank.last_kernel, shape, D, layout, inplace)
      m = x.shape[ax]
      r = ref[at:at + m] if tm else ref[:, at:at + m]
      assert same_bits(y.cpu().numpy(), r), (shape, D, layout, inplace, at)
      at += m


@pytest.mark.parametrize("D", [16, 63, 64, 65, 109, 128, 257, 441, 512, 513, 1000, 3000])
def test_one_string_lanes_over_the_delay(alz, oracle, D):
  """A single Karplus-Strong string (lazy_synth.py:624-657: comb.tau(...).linearize()(zeros(), memory=white_noise)): one
  channel, lanes over the delay line (k_comb_cm), as [1, N] and as the reference's [N, 1] column, from a noise-filled
  delay line, in place and not; bit-exact, block after block."""
  import torch
  rng = np.random.default_rng(D)
  b, a = _comb_case("lin", D, 1, rng)
  na = a.shape[1]
  yh0 = rng.uniform(-1, 1, (1, na - 1))
  lens = [5 * D + 10, 700, 66, 2 * D, 256 * 3]
  xs = [np.zeros(lens[0]), rng.uniform(-1e-3, 1e-3, lens[1]), np.zeros(lens[2]), rng.uniform(-1, 1, lens[3]), np.zeros(lens[4])]
  ref = oracle.bank([1], [na], b, a, np.concatenate(xs)[None, :], layout="chan", yh=yh0.copy())[0]
  for layout in ("chan", "time"):
    for inplace in (False, True):
      bank = alz.FilterBank([(b, a)], n_inputs=1)
      bank.set_state(np.zeros((1, 1)), yh0)
      at = 0
      for x in xs:
        xd = torch.from_numpy(x[None, :] if layout == "chan" else x[:, None]).cuda()
        y = bank.process(xd, layout=layout, out=xd if inplace else None)
        assert bank.last_kernel == ("k_string" if D <= 512 else "k_comb_cm"), bank.last_kernel
        assert same_bits(y.cpu().numpy().ravel(), ref[at:at + len(x)]), (D, layout, inplace, at)
        at += len(x)


@pytest.mark.parametrize("C", [2, 5, 17, 40])
@pytest.mark.parametrize("shape,D", [("fb", 24), ("lin", 109), ("lin", 441), ("fb", 700), ("ff", 300), ("mixed", 109)])
def test_comb_on_a_few_channels_of_time_major_rows(alz, oracle, C, shape, D):
  """Stereo rows [N, 2] (and other channel counts that are not whole groups of 16, up to 64) of a time-major block: the
  wave-per-channel kernels with strided loads and stores -- lanes over the delay line -- instead of round 1's lane per channel;
  bit-exact across blocks, in place where the numerator is b0 alone."""
  import torch
  rng = np.random.default_rng(C * 1000 + D)
  b, a = _comb_case(shape, D, C, rng)
  nb, na = b.shape[1], a.shape[1]
  lens = [3 * D + 37, D // 2 + 3, 600, 256, 1]
  xs = [rng.uniform(-1, 1, (m, C)) for m in lens]
  xh0 = rng.uniform(-1, 1, (C, max(nb - 1, 1)))
  yh0 = rng.uniform(-1, 1, (C, max(na - 1, 1)))
  ref = oracle.bank([nb], [na], b, a, np.concatenate(xs), layout="time", xh=xh0.copy(), yh=yh0.copy())
  want = "k_string" if (D <= 512 and shape in ("fb", "lin")) else "k_comb_cm"
  for inplace in ((False, True) if nb == 1 else (False,)):
    bank = alz.FilterBank([(b, a)], n_inputs=C)
    bank.set_state(xh0, yh0)
    at = 0
    for x in xs:
      xd = torch.from_numpy(x).cuda()
      y = bank.process(xd, layout="time", out=xd if inplace else None)
      assert bank.last_kernel == want, (bank.last_kernel, want)
      assert same_bits(y.cpu().numpy(), ref[at:at + x.shape[0]]), (shape, D, C, inplace, at)
      at += x.shape[0]


def test_comb_outer_bank_and_wide_block(alz, oracle):
  """A comb per (set, input) pair -- an OUTER bank reading by input index -- and a block wide and long enough for every
  workgroup / wave to run many steps (4096 channels x 4096 samples, D = 441: bench.py's comb_fb shape, shortened)."""
  import torch
  rng = np.random.default_rng(5)
  D, S, B = 120, 32, 3
  g = rng.uniform(0.5, 0.95, B)
  a = np.zeros((B, D + 1)); a[:, 0] = 1; a[:, D] = -g
  b = np.ones((B, 1))
  for layout in ("time", "chan"):
    x = rng.uniform(-1, 1, (900, S) if layout == "time" else (S, 900))
    bank = alz.FilterBank([(b, a)], n_inputs=S, mode="outer")
    bank.reset()
    y = bank.process(torch.from_numpy(x).cuda(), layout=layout).cpu().numpy()
    assert bank.last_kernel in ("k_comb_tm", "k_comb_cm", "k_string") or "k_expand" in bank.last_kernel, bank.last_kernel
    for s in range(B):
      xs = x if layout == "time" else x
      r = oracle.bank([1], [D + 1], b[s], a[s], x, layout=layout)
      got = y[:, s * S:(s + 1) * S] if layout == "time" else y[s * S:(s + 1) * S]
      assert same_bits(got, r), (layout, s)
  import bench
  C, N, D = 4096, 4096, 441
  bb, aa = bench.comb_coefs(C, D)
  x = rng.uniform(-1, 1, (N, C))
  ref = oracle.bank([1], [D + 1], bb, aa, x, layout="time")
  bank = alz.FilterBank([(bb, aa)], n_inputs=C)
  bank.reset()
  assert same_bits(bank.process(torch.from_numpy(x).cuda(), layout="time").cpu().numpy(), ref) and bank.last_kernel == "k_comb_tm"
  xt = np.ascontiguousarray(x.T)
  bank.reset()
  assert same_bits(bank.process(torch.from_numpy(xt).cuda(), layout="chan").cpu().numpy(), ref.T) and bank.last_kernel == "k_comb_cm"


def test_random_structure_sweep(alz, oracle):
  """Seeded sweep over filter structures: random orders, random zero taps (uniform or per
  channel), gains, cascades of 1-3 sections, both layouts, ragged sizes -- whatever kernel the
  dispatcher picks, the result must be the oracle's, bit for bit."""
  rng = np.random.default_rng(20260924)
  kernels = set()
  for trial in range(60):
    C = int(rng.choice([1, 5, 16, 33, 64, 80, 130]))
    N = int(rng.choice([1, 9, 64, 65, 200, 517]))
    nsec = int(rng.choice([1, 1, 2, 3]))
    layout = "time" if rng.random() < 0.6 else "chan"
    per_channel = rng.random() < 0.6
    secs, nbs, nas = [], [], []
    for _ in range(nsec):
      nb = int(rng.choice([1, 2, 3, 3, 5, 9, 20]))
      na = int(rng.choice([1, 2, 3, 3, 3, 4, 7]))
      rows = C if per_channel else 1
      b = rng.uniform(-1, 1, (rows, nb))
      a = np.concatenate([np.ones((rows, 1)), rng.uniform(-1, 1, (rows, na - 1)) * (0.6 / max(na - 1, 1))], axis=1)
      if rng.random() < 0.5:                       # zero a whole tap column (uniform pattern)
        b[:, rng.integers(nb)] = 0.0
      if rng.random() < 0.3 and per_channel:       # zero taps on some channels only (mixed pattern)
        b[::2, rng.integers(nb)] = 0.0
      if na > 1 and rng.random() < 0.3:
        a[:, 1 + rng.integers(na - 1)] = 0.0
      if rng.random() < 0.25:
        a[:, 0] = rng.uniform(0.5, 2.0, rows) * rng.choice([-1.0, 1.0])
      if not np.any(b) and not np.any(a[:, 1:]):
        b[:, 0] = 0.5
      secs.append((b if per_channel else b[0], a if per_channel else a[0]))
      nbs.append(nb); nas.append(na)
    zero = float(rng.choice([0.0, 0.0, 0.25]))
    mem = None if rng.random() < 0.5 else rng.uniform(-1, 1, 3).tolist()
    x = rng.uniform(-1, 1, (N, C) if layout == "time" else (C, N))
    bank = alz.FilterBank(secs, n_inputs=C)
    bank.reset(memory=mem, zero=zero)
    y = bank.process(x, layout=layout)
    kernels.update(bank.last_kernel.split("+"))
    bcat = np.concatenate([np.atleast_2d(s[0]) for s in secs], axis=1)
    acat = np.concatenate([np.atleast_2d(s[1]) for s in secs], axis=1)
    if not per_channel:
      bcat, acat = bcat[0], acat[0]
    xh = np.full((C, max(sum(n - 1 for n in nbs), 1)), zero)
    yh = np.concatenate([np.array(alz.memory_to_hist(mem, n - 1, zero), dtype=float) for n in nas])
    yh = np.tile(yh, (C, 1)) if yh.size else np.zeros((C, 1))
    ref = oracle.bank(nbs, nas, bcat, acat, x, layout=layout, xh=xh, yh=np.ascontiguousarray(yh), zero=zero)
    assert same_bits(y, ref), (trial, C, N, nbs, nas, layout, per_channel, bank.last_kernel)
  assert len(kernels) >= 4, kernels      # the sweep really exercised several kernel families


def test_fused_mode_is_opt_in_and_within_contract(alz, oracle):
  """alz_bank_set_fused: FMA contraction in the streaming kernel.  Floating-point parity only:
  per channel max|y - y_ref| / max|y_ref| <= 1e-9 (the north star's contract is 1e-6); the
  default mode stays bit-exact."""
  rng = np.random.default_rng(31)
  C, N = 256, 8192
  b, a = resonator_bank(C)
  x = rng.uniform(-1, 1, (N, C))
  ref = oracle.bank([3], [3], b, a, x)
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  bank.reset()
  assert same_bits(bank.process(x), ref)                 # default: exact
  bank.set_fused(True)
  bank.reset()
  y = bank.process(x)
  assert "fma" in bank.last_kernel
  nerr = (np.abs(y - ref).max(axis=0) / np.abs(ref).max(axis=0)).max()
  assert nerr <= 1e-9
  bank.set_fused(False)
  bank.reset()
  assert same_bits(bank.process(x), ref)


@pytest.mark.parametrize("layout", ["time", "chan"])
def test_outer_bank_streaming_kernels(alz, oracle, layout):
  """A single-section filterbank (every resonator on every stream) and a 3-section OUTER cascade
  (not one of the fused shapes): the streaming kernels take OUTER banks when the stream count is
  a multiple of the group width."""
  rng = np.random.default_rng(8)
  S, B, N = 48, 7, 1000
  x = rng.uniform(-1, 1, (N, S))
  xin = x if layout == "time" else np.ascontiguousarray(x.T)
  bb, aa = resonator_bank(B)
  fb = alz.FilterBank([(bb, aa)], n_inputs=S, mode="outer")
  fb.reset()
  y = fb.process(xin, layout=layout)
  assert "k_duo" in fb.last_kernel or "k_wave" in fb.last_kernel
  y = y if layout == "time" else y.T
  for band in range(B):
    assert same_bits(y[:, band * S:(band + 1) * S], oracle.bank([3], [3], bb[band], aa[band], x))
  secs = [(bb, aa), (bb[:, :1] * 2.0, aa[:, :2]), (bb, aa)]
  fb3 = alz.FilterBank(secs, n_inputs=S, mode="outer")
  fb3.reset()
  y3 = fb3.process(xin, layout=layout)
  y3 = y3 if layout == "time" else y3.T
  for band in (0, B - 1):
    ref = oracle.bank([3, 1, 3], [3, 2, 3], np.concatenate([bb[band], bb[band, :1] * 2.0, bb[band]]),
                      np.concatenate([aa[band], aa[band, :2], aa[band]]), x)
    assert same_bits(y3[:, band * S:(band + 1) * S], ref)


def test_fir_channel_major_per_channel_taps_long_block(alz, oracle):
  rng = np.random.default_rng(404)
  C, N, nb = 37, 5000, 100                      # several 2048-output runs per channel, ragged end
  b = rng.uniform(-1, 1, (C, nb))
  b[::4, 7] = 0.0
  b[3] = 0.0                                    # all-zero channel
  a = rng.uniform(0.5, 2.0, (C, 1))
  x = rng.uniform(-1, 1, (C, N))
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  bank.reset(zero=0.5)
  y = bank.process(x, layout="chan")
  assert bank.last_kernel == "k_fir_cm"
  ref = oracle.bank([nb], [1], b, a, x, layout="chan", xh=np.full((C, nb - 1), 0.5), zero=0.5)
  assert same_bits(y, ref)
  assert np.all(y[3] == 0.5)


@pytest.mark.parametrize("layout", ["time", "chan"])
def test_streaming_with_gain_division(alz, oracle, layout):
  """a0 != 1 on a biquad bank large enough for the streaming kernel: (sum) / a0 per sample
  (lazy_filters.py:236-240), bit-exact, including a0 == -1, a0 == 1 rows and block splits."""
  rng = np.random.default_rng(31)
  C, N = 256, 1000
  b = rng.uniform(-1, 1, (C, 3))
  a = rng.uniform(-.45, .45, (C, 3))
  a[:, 0] = rng.choice([2., .5, -1., 1., 3.7], C)
  x = rng.uniform(-1, 1, (N, C) if layout == "time" else (C, N))
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  bank.reset(zero=.125)
  cut = 448
  parts = [bank.process(np.ascontiguousarray(x[:cut] if layout == "time" else x[:, :cut]), layout=layout)]
  assert "k_duo<16,div>" in bank.last_kernel
  parts.append(bank.process(np.ascontiguousarray(x[cut:] if layout == "time" else x[:, cut:]), layout=layout))
  got = np.concatenate(parts, axis=0 if layout == "time" else 1)
  ref = oracle.bank([3], [3], b, a, x, layout=layout, zero=.125)
  assert np.array_equal(got.view(np.uint64), ref.view(np.uint64))


@pytest.mark.gpu
def test_fir_narrow_and_long(alz, oracle):
  """Shared-tap FIR on 16 channels x 3.4 M samples: more runs than the grid's y range allows blocks (the launcher
  clamps the grid and lets every block take several runs)."""
  import torch
  rng = np.random.default_rng(5)
  C, n, nb = 16, 3_400_000, 24
  b = np.tile(rng.uniform(-1, 1, nb), (C, 1))
  a = np.ones((C, 1))
  x = rng.uniform(-1, 1, (n, C))
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  y = bank.process(torch.from_numpy(x).cuda(), layout="time").cpu().numpy()
  assert "k_fir" in bank.last_kernel, bank.last_kernel
  # the WHOLE block, head to tail, bit for bit (round-3 review: the tail was only checked for finiteness)
  ref = oracle.bank([nb], [1], b, a, x, layout="time")
  assert y.shape == ref.shape
  same = y.view(np.uint64) == ref.view(np.uint64)
  assert same.all(), "first differing row %d of %d" % (int(np.argmin(same.all(axis=1))), n)


@pytest.mark.parametrize("layout", ["time", "chan"])
def test_row_padded_torch_tensors(alz, oracle, layout):
  """``process`` takes 2-D CUDA tensors whose rows are contiguous but padded (a column-sliced view of a wider
  allocation): the C ABI's ldx / ldy.  Round 5 measured what the pitch is worth -- channel-major blocks whose rows are a
  large power of two apart (2^18 .. 2^20 doubles) land consecutive rows on the same HBM channel and gain 3 - 16 % from 32
  doubles of padding (profiles/r05_pitch_probe.log) -- so a caller must be able to choose it.  Same doubles as the
  contiguous call; a view whose rows are not contiguous is refused."""
  import torch
  C, n = 256, 4096 + 64
  rng = np.random.default_rng(31)
  r, th = rng.uniform(.8, .999, C), rng.uniform(.02, 3., C)
  b = rng.uniform(-1, 1, (C, 3))
  a = np.stack([np.ones(C), -2 * r * np.cos(th), r * r], axis=1)
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  x = rng.uniform(-1, 1, (n, C) if layout == "time" else (C, n))
  ref = oracle.bank([3], [3], b, a, x, layout=layout)
  rows, cols = x.shape
  for pad_x, pad_y in ((0, 32), (18, 0), (34, 34)):
    xw = torch.zeros((rows, cols + pad_x), dtype=torch.float64, device="cuda")
    yw = torch.full((rows, cols + pad_y), -7.0, dtype=torch.float64, device="cuda")
    xv, yv = xw[:, :cols], yw[:, :cols]
    xv.copy_(torch.from_numpy(x))
    bank.reset()
    out = bank.process(xv, layout=layout, out=yv)
    assert out.data_ptr() == yw.data_ptr()
    assert same_bits(yv.cpu().numpy(), ref), (layout, pad_x, pad_y, bank.last_kernel)
    if pad_y:
      assert bool((yw[:, cols:] == -7.0).all())            # the padding is not written
  with pytest.raises(ValueError):
    bank.process(torch.zeros((rows, 2 * cols), dtype=torch.float64, device="cuda")[:, ::2], layout=layout)
