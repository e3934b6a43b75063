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


# ---------------------------------------------------------------------------------------------------
# Differential fuzz of alz_tv_process_dev through the raw C ABI: every curated presence pattern (k_tvp),
# other shapes (k_tv), every way a tap can get its values (constant, series shared by the bank --
# contiguous or strided --, series per channel, series the host already negated), the three gain
# modes, batch-ragged lengths, both layouts; against the pure-Python restatement, bit for bit.
# ---------------------------------------------------------------------------------------------------
PATTERNS = [(1, 1), (3, 1), (1, 3), (3, 3), (5, 3), (7, 3), (1, 2), (1, 0), (2, 0), (3, 0), (4, 0), (7, 0),
            (6, 3), (5, 1)]                       # the last two are not curated: the general kernel


@pytest.mark.parametrize("case", range(56))
def test_raw_abi_fuzz(al, case):
  import ctypes
  import torch
  from audiolazy_amd import _ffi
  rng = np.random.default_rng(1000 + case)
  pb, pa = PATTERNS[case % len(PATTERNS)]
  N = int(rng.choice([1, 5, 15, 16, 17, 31, 32, 50, 100]))
  C = int(rng.choice([2, 3, 64, 70]))
  layout = "time" if rng.random() < .6 else "chan"
  gain = float(rng.choice([1.0, -1.0, 2.5]))
  nb = max(k + 1 for k in range(3) if (pb >> k) & 1)
  na = 1 + max([k for k in (1, 2) if (pa >> (k - 1)) & 1], default=0)
  L = _ffi.load()
  keep, ref_b, ref_a = [], [], []
  x = rng.uniform(-1, 1, (N, C))

  def make_tap(present, is_a):
    """-> (TvTap, reference value: float or [N] or [N, C] array of what the tap IS, sign as the reference sees it)"""
    if not present:
      return _ffi.TvTap(0.0, None, 0, 0, 0), 0.0
    how = rng.choice(["const", "shared", "strided", "lane", "negated"] if is_a else ["const", "shared", "strided", "lane"])
    if how == "const":
      v = float(rng.uniform(-.9, .9)) or .5
      return _ffi.TvTap(v, None, 0, 0, 0), v
    if how in ("shared", "negated"):
      vals = rng.uniform(-.9, .9, N)
      held = -vals if how == "negated" else vals           # what the device buffer holds
      t = torch.from_numpy(held.copy()).cuda(); keep.append(t)
      return _ffi.TvTap(0.0, t.data_ptr(), 1, 0, _ffi.TV_NEGATED if how == "negated" else 0), vals
    if how == "strided":
      vals = rng.uniform(-.9, .9, N)
      buf = np.zeros(3 * N); buf[::3] = vals
      t = torch.from_numpy(buf).cuda(); keep.append(t)
      return _ffi.TvTap(0.0, t.data_ptr(), 3, 0, 0), vals
    vals = rng.uniform(-.9, .9, (N, C))
    arr = vals if layout == "time" else vals.T
    t = torch.from_numpy(np.ascontiguousarray(arr)).cuda(); keep.append(t)
    return _ffi.TvTap(0.0, t.data_ptr(), *((C, 1) if layout == "time" else (1, N)), 0), vals

  tb = (_ffi.TvTap * nb)()
  for k in range(nb):
    tb[k], v = make_tap((pb >> k) & 1, False); ref_b.append(v)
  ta = (_ffi.TvTap * na)()
  ta[0] = _ffi.TvTap(gain, None, 0, 0, 0); ref_a.append(gain)
  for k in range(1, na):
    ta[k], v = make_tap((pa >> (k - 1)) & 1, True); ref_a.append(v)
  xd = torch.from_numpy(np.ascontiguousarray(x if layout == "time" else x.T)).cuda()
  y = torch.empty_like(xd)
  xh = torch.full((max(nb - 1, 1), C), .125, dtype=torch.float64, device="cuda")
  yh = torch.full((max(na - 1, 1), C), .125, dtype=torch.float64, device="cuda")
  lay = _ffi.TIME_MAJOR if layout == "time" else _ffi.CHAN_MAJOR
  ld = C if layout == "time" else N
  stream = torch.cuda.current_stream().cuda_stream
  _ffi.check(L.alz_tv_process_dev(nb, ctypes.cast(tb, ctypes.c_void_p), na, ctypes.cast(ta, ctypes.c_void_p), C,
                                  xd.data_ptr(), y.data_ptr(), N, lay, ld, ld, xh.data_ptr(), yh.data_ptr(),
                                  .125, 0, ctypes.c_void_p(stream)))
  torch.cuda.synchronize()
  got = y.cpu().numpy()
  got = got if layout == "time" else got.T
  for ch in sorted(set([0, 1, C // 2, C - 1])):
    pick = lambda v: v[:, ch] if isinstance(v, np.ndarray) and v.ndim == 2 else v
    ref = oracle.tv_df1([pick(v) for v in ref_b], [pick(v) for v in ref_a], x[:, ch], memory=[.125] * (na - 1), zero=.125)
    assert same_bits(got[:, ch], ref), (case, pb, pa, N, C, layout, gain, ch)


# ---------------------------------------------------------------------------------------------------
# k_tvduo (csrc/alz_tvduo.hip): a bank whose coefficient series are shared by the channels -- full
# 64-row tiles on the two-wave streaming kernel, the ragged rest on the lane-per-channel kernels.
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pattern", [(1, 1), (3, 1), (1, 3), (3, 3), (5, 3), (7, 3), (1, 2)])
@pytest.mark.parametrize("C", [16, 256])
@pytest.mark.parametrize("layout", ["time", "chan"])
def test_bank_steered_by_shared_series(al, pattern, C, layout):
  import torch
  from audiolazy_amd import timevar
  pb, pa = pattern
  rng = np.random.default_rng(100 * pb + 10 * pa + C)
  N1, N2 = 64 * 5 + 17, 64 * 3              # two blocks: tiles + ragged tail, then the stream goes on
  N = N1 + N2
  x = rng.uniform(-1, 1, (N, C))
  nb = max(k + 1 for k in range(3) if (pb >> k) & 1)
  na = 1 + max(k for k in (1, 2) if (pa >> (k - 1)) & 1)
  kinds = ["series", "const", "series"]      # b0 b1 b2: a constant tap next to series taps
  b_ref, a_ref = [], [1.]
  for k in range(nb):
    if not (pb >> k) & 1:
      b_ref.append(0.)
    elif kinds[k] == "const":
      b_ref.append(.37)
    else:
      b_ref.append(rng.uniform(-.9, .9, N))
  for k in range(1, na):
    a_ref.append(rng.uniform(-.45, .45, N) if (pa >> (k - 1)) & 1 else 0.)
  dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
  xh = torch.full((max(nb - 1, 1), C), .25, dtype=torch.float64, device="cuda")
  yh = torch.full((max(na - 1, 1), C), -.125, dtype=torch.float64, device="cuda")
  xh2, yh2 = xh.clone(), yh.clone()

  def run(lo, hi, per_channel, xh_, yh_):
    def tap(v):
      if not isinstance(v, np.ndarray):
        return v
      if not per_channel:
        return dev(v[lo:hi])
      full = np.repeat(v[lo:hi, None], C, axis=1)
      return dev(full if layout == "time" else full.T)
    xb = x[lo:hi] if layout == "time" else x[lo:hi].T
    out = timevar.process_block([tap(v) for v in b_ref], [tap(v) for v in a_ref], dev(xb),
                                xh=xh_, yh=yh_, zero=.25, layout=layout).cpu().numpy()
    return out if layout == "time" else out.T
  y = np.concatenate([run(0, N1, False, xh, yh), run(N1, N, False, xh, yh)])
  # the same values as per-channel series: the lane-per-channel kernel, which must give the same doubles
  y_lane = np.concatenate([run(0, N1, True, xh2, yh2), run(N1, N, True, xh2, yh2)])
  assert same_bits(y, y_lane)
  assert torch.equal(xh, xh2) and torch.equal(yh, yh2)
  for ch in (0, C // 2 + 1, C - 1):
    ref = oracle.tv_df1(b_ref, a_ref, x[:, ch], memory=[-.125] * (na - 1), zero=.25)
    assert same_bits(y[:, ch], ref), (pattern, ch)


# ---------------------------------------------------------------------------------------------------
# k_tvpc (csrc/alz_tvpc.hip): per-channel coefficient series on time-major rows -- full 64-row tiles on the
# three-wave kernel, the ragged rest on the lane-per-channel kernels; every channel against the oracle.
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pattern,series_taps", [((5, 3), "b0 a1 a2"), ((1, 3), "b0 a1 a2"), ((5, 3), "b0 b2 a1"), ((3, 3), "a1 a2"),
                                                   ((7, 3), "b1"), ((1, 1), "b0 a1"), ((3, 1), "a1"), ((1, 3), "a2")])
@pytest.mark.parametrize("C", [16, 80])
def test_bank_with_per_channel_series(al, pattern, series_taps, C):
  import torch
  from audiolazy_amd import timevar
  pb, pa = pattern
  rng = np.random.default_rng(1000 * pb + 100 * pa + C + len(series_taps))
  N1, N2 = 64 * 6 + 9, 64 * 4               # tiles + ragged tail, then the stream goes on
  N = N1 + N2
  x = rng.uniform(-1, 1, (N, C))
  nb = max(k + 1 for k in range(3) if (pb >> k) & 1)
  na = 1 + max(k for k in (1, 2) if (pa >> (k - 1)) & 1)
  b_ref = [(rng.uniform(-.9, .9, (N, C)) if "b%d" % k in series_taps else .37 - .1 * k) if (pb >> k) & 1 else 0. for k in range(nb)]
  a_ref = [1.] + [(rng.uniform(-.45, .45, (N, C)) if "a%d" % k in series_taps else .2 * k - .3) if (pa >> (k - 1)) & 1 else 0.
                  for k in range(1, na)]
  dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
  xh = torch.full((max(nb - 1, 1), C), .25, dtype=torch.float64, device="cuda")
  yh = torch.full((max(na - 1, 1), C), -.125, dtype=torch.float64, device="cuda")
  tap = lambda v, lo, hi: dev(v[lo:hi]) if isinstance(v, np.ndarray) else v
  run = lambda lo, hi: timevar.process_block([tap(v, lo, hi) for v in b_ref], [tap(v, lo, hi) for v in a_ref], dev(x[lo:hi]),
                                             xh=xh, yh=yh, zero=.25).cpu().numpy()
  y = np.concatenate([run(0, N1), run(N1, N)])
  pick = lambda v, ch: v[:, ch] if isinstance(v, np.ndarray) else v
  for ch in range(C):
    ref = oracle.tv_df1([pick(v, ch) for v in b_ref], [pick(v, ch) for v in a_ref], x[:, ch], memory=[-.125] * (na - 1), zero=.25)
    assert same_bits(y[:, ch], ref), (pattern, series_taps, ch)


def test_rows_idiom_with_per_channel_coefficient_streams(al):
  """The reference's vector-valued idiom with coefficient Streams whose items are rows (one coefficient per channel):
  ``(Stream(rows_b0) + ...) / (1 + Stream(rows_a1) z^-1 + ...)`` called on rows -- denominators negated on the host
  (ALZ_TV_NEGATED), 16-channel groups on the three-wave kernel."""
  C, N = 32, 64 * 7 + 5
  rng = np.random.default_rng(42)
  x = rng.uniform(-1, 1, (N, C))
  b0, a1, a2 = rng.uniform(-.9, .9, (N, C)), rng.uniform(-.45, .45, (N, C)), rng.uniform(-.3, .3, (N, C))
  filt = (al.Stream(iter(b0)) - .5 * al.z ** -2) / (1 + al.Stream(iter(a1)) * al.z ** -1 + al.Stream(iter(a2)) * al.z ** -2)
  got = np.array(list(filt(iter(x), zero=np.zeros(C))))
  for ch in range(C):
    ref = oracle.tv_df1([b0[:, ch], 0., -.5], [1., a1[:, ch], a2[:, ch]], x[:, ch])
    assert same_bits(got[:, ch], ref), ch
