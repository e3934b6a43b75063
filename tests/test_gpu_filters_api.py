"""The reference's operator surface on the GPU engine: these read like the reference's own
tests / doctests (cited per case) with `audiolazy_amd` in place of `audiolazy`.
All comparisons with reference-generated vectors are bit-exact."""
import numpy as np
import pytest

from conftest import load_golden, unhex

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def al():
  import audiolazy_amd
  assert audiolazy_amd.device_count() >= 1
  return audiolazy_amd


def same_bits(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return a.shape == b.shape and bool(np.all(a.view(np.uint64) == b.view(np.uint64)))


def test_doctest_known_answers(al):
  z, ZFilter = al.z, al.ZFilter
  filt = (1 + z ** -1) / (1 - z ** -1)                       # lazy_filters.py:722-726
  assert list(filt([1, 5, -4, -7, 9])) == [1.0, 7.0, 8.0, -3.0, -1.0]
  filt = ZFilter([1, 1], [1, -1])                            # lazy_filters.py:735-742
  result = list(filt([1, 5, -4, -7, 9], memory=[3], zero=0))
  assert result == [4, 10, 11, 0, 2]
  assert list((filt * z ** -1)(result, zero=0)) == [0, 4, 18, 39, 50]
  acc = 1 / (1 - z ** -1)                                    # audiolazy/__init__.py:30-33
  assert list(acc([1, 3, 2, 1, 3, 2, 1, 3])) == [1, 4, 6, 7, 10, 12, 13, 16]
  data = [.1, .2, .4, .3, .2, -.1, -.3, -.2]                 # README.rst:286-300
  np.testing.assert_allclose(list((1 - z ** -1)(data)), [.1, .1, .2, -.1, -.1, -.3, -.2, .1], atol=1e-15)


def test_errors(al):
  z, ZFilter = al.z, al.ZFilter
  with pytest.raises(ValueError):                            # lazy_filters.py:165-168
    (z ** 2)([1, 2, 3])
  with pytest.raises(ValueError):       # the common delay is cancelled first (:126-132): 1 / z ** -1 is z
    ZFilter([1.], [0., 1.])([1, 2, 3])
  with pytest.raises(ZeroDivisionError):                     # lazy_filters.py:177-178, at the engine's gate
    al.FilterBank([([1.], [0., 1.])], n_inputs=1)


def test_designs_called_like_the_reference(al):
  case_by_tag = {c["tag"]: c for c in load_golden("filters.json")["cases"]}
  x = unhex(load_golden("filters.json")["x"])
  s, Hz = al.sHz(48000)
  for tag, filt in [("lowpass.pole**2@1k", al.lowpass.pole(1000 * Hz) ** 2),
                    ("highpass.z@1k", al.highpass.z(1000 * Hz)),
                    ("resonator.z_exp@1k/100", al.resonator.z_exp(1000 * Hz, 100 * Hz)),
                    ("resonator.freq_poles_exp@440/30", al.resonator.freq_poles_exp(440 * Hz, 30 * Hz)),
                    ("comb.fb(5,.5)", al.comb.fb(5, .5)),
                    ("comb.tau(20,100)", al.comb.tau(20, 100))]:
    c = case_by_tag[tag]
    assert same_bits(list(filt(x[:c["x_len"]])), unhex(c["y"])), tag


def test_cascade_parallel_containers(al):
  s, Hz = al.sHz(48000)
  fs = [al.lowpass.pole(800 * Hz), al.highpass.z(200 * Hz), al.resonator.z_exp(1500 * Hz, 80 * Hz)]
  cases = load_golden("containers.json")
  x = unhex(cases[0]["x"])
  assert same_bits(list(al.CascadeFilter(fs)(x)), unhex(cases[0]["y"]))     # lazy_filters.py:988-990
  fs2 = [al.resonator.poles_exp(700 * Hz, 50 * Hz), al.resonator.z_exp(900 * Hz, 60 * Hz)]
  y = list(al.CascadeFilter(fs2)(unhex(cases[1]["x"]), memory=[0.3, -0.2], zero=0.1))
  assert same_bits(y, unhex(cases[1]["y"]))
  assert same_bits(list(al.ParallelFilter(fs)(x)), unhex(cases[2]["y"]))    # lazy_filters.py:1048-1054
  assert list(al.ParallelFilter()([1, 2, 3], zero=.5)) == [.5, .5, .5]      # :1049-1051
  assert list(al.CascadeFilter()([1, 2, 3])) == [1, 2, 3]


def test_gammatone_bands_and_bank(al):
  aud = load_golden("auditory.json")
  x = unhex(aud["x"])
  for g in aud["gammatone"]:
    band = getattr(al.gammatone, g["strategy"])(unhex(g["freq"]), unhex(g["bw"]))
    assert same_bits(list(band(x)), unhex(g["y"])), g["strategy"]
  # a bank = every band on every stream, one launch set: [N, B*S], channel = band*S + stream
  s, Hz = al.sHz(48000)
  fcs = [f * Hz for f in al.erb_space(50., 20000., 6)]
  rng = np.random.default_rng(3)
  S, N = 5, 777
  xs = rng.uniform(-1, 1, (N, S))
  for strat in ("slaney", "klapuri", "sampled"):
    bank = al.gammatone_bank(fcs, S, strategy=strat, Hz=Hz)
    bank.reset()
    y = bank.process(xs)
    assert y.shape == (N, 6 * S)
    k = al.gammatone_erb_constants(4)[0]
    for b, fc in enumerate(fcs):
      band = getattr(al.gammatone, strat)(fc, k * al.erb(fc, Hz))
      for st in (0, S - 1):
        assert same_bits(y[:, b * S + st], list(band(xs[:, st].tolist()))), (strat, b)


def test_stream_blocks_feed(al):
  # Stream.blocks is the feed of the blocked engine (lazy_stream.py:215-220)
  s, Hz = al.sHz(48000)
  filt = al.lowpass.pole(1000 * Hz) ** 2
  x = unhex(load_golden("filters.json")["x"])
  ref = unhex(load_golden("filters.json")["cases"][0]["y"])
  blks = [list(b) for b in filt(x).blocks(size=64)]
  assert len(blks) == (len(x) + 63) // 64
  flat = [v for b in blks for v in b][:len(x)]
  assert same_bits(flat, ref)
  # small pull blocks give the same stream
  bank = al.FilterBank([(filt.numlist, filt.denlist)], n_inputs=1)
  assert same_bits(list(bank(x, block=7)), ref)


def test_karplus_strong_style_comb_with_callable_memory(al):
  # lazy_synth.py:624-657: comb.tau(...)(zeros(), memory=white_noise) -- memory as a callable (:188-189)
  import random
  random.seed(8)
  noise = [random.uniform(-1, 1) for _ in range(50)]
  filt = al.comb.tau(50, 2000)
  y = list(filt([0.] * 400, memory=lambda n: noise[:n]))
  from oracle import oracle
  ref = oracle.df1(filt.numlist, filt.denlist, [0.] * 400, memory=noise)
  assert same_bits(y, ref)


@pytest.mark.parametrize("layout", ["chan", "time"])
@pytest.mark.parametrize("strategy", ["slaney", "klapuri", "sampled"])
def test_fused_gammatone_bank(al, strategy, layout):
  """cfg4 shape: x [S, N] -> y [B, S, N] (or time-major [N, B*S]); the fused cascade kernel takes
  the full 16-sample tiles of the 64-stream groups, the ragged tail goes section by section."""
  from oracle import oracle
  s, Hz = al.sHz(48000)
  B, S, N = 5, 128, 1000 + 7
  fcs = [f * Hz for f in al.erb_space(50., 20000., B)]
  rng = np.random.default_rng(11)
  x = rng.uniform(-1, 1, (S, N))
  bank = al.gammatone_bank(fcs, S, strategy=strategy, Hz=Hz)
  bank.reset()
  y = bank.process(x if layout == "chan" else np.ascontiguousarray(x.T), layout=layout)
  assert any(k in bank.last_kernel for k in ("k_casc", "k_pipe", "k_flow"))
  if layout == "time":
    y = y.T
  assert y.shape == (B * S, N)
  k = al.gammatone_erb_constants(4)[0]
  for b, fc in enumerate(fcs):
    band = getattr(al.gammatone, strategy)(fc, k * al.erb(fc, Hz))
    secs = [(f.numlist, f.denlist) for f in band]
    nb, na = [len(q[0]) for q in secs], [len(q[1]) for q in secs]
    ref = oracle.bank(nb, na, np.concatenate([q[0] for q in secs]), np.concatenate([q[1] for q in secs]),
                      x, layout="chan")
    assert same_bits(y[b * S:(b + 1) * S], ref), (strategy, b)
  # the next block continues every cascade from the state the fused kernel left
  x2 = rng.uniform(-1, 1, (S, 100))
  y2 = bank.process(x2 if layout == "chan" else np.ascontiguousarray(x2.T), layout=layout)
  y2 = y2.T if layout == "time" else y2
  band = getattr(al.gammatone, strategy)(fcs[2], k * al.erb(fcs[2], Hz))
  secs = [(f.numlist, f.denlist) for f in band]
  whole = oracle.bank([len(q[0]) for q in secs], [len(q[1]) for q in secs],
                      np.concatenate([q[0] for q in secs]), np.concatenate([q[1] for q in secs]),
                      np.concatenate([x, x2], axis=1), layout="chan")
  assert same_bits(y2[2 * S:3 * S], whole[:, N:])


def test_fused_diagonal_cascade_of_biquads(al):
  from oracle import oracle
  rng = np.random.default_rng(2)
  C, N = 192, 500
  b1, b2 = rng.uniform(-1, 1, (C, 3)), rng.uniform(-1, 1, (C, 3))
  a1 = np.concatenate([np.ones((C, 1)), rng.uniform(-.4, .4, (C, 2))], axis=1)
  a2 = np.concatenate([np.ones((C, 1)), rng.uniform(-.4, .4, (C, 2))], axis=1)
  x = rng.uniform(-1, 1, (N, C))
  bank = al.FilterBank([(b1, a1), (b2, a2)], n_inputs=C)
  bank.reset(memory=[0.25, -0.5], zero=0.125)
  y = bank.process(x)
  assert any(k in bank.last_kernel for k in ("k_casc", "k_pipe", "k_flow"))
  yh = np.tile(np.array([0.25, -0.5, 0.25, -0.5]), (C, 1))
  ref = oracle.bank([3, 3], [3, 3], np.concatenate([b1, b2], axis=1), np.concatenate([a1, a2], axis=1), x,
                    xh=np.full((C, 4), 0.125), yh=yh, zero=0.125)
  assert same_bits(y, ref)


def test_karplus_strong_golden(al):
  """lazy_synth.py:624-657 end to end: seeded white_noise memory (a callable), fractional-period
  comb linearized to two feedback taps, endless input of zeros -- bit-exact with the reference."""
  import random
  for case in load_golden("karplus.json"):
    random.seed(case["seed"])
    ks = al.karplus_strong(unhex(case["freq"]), unhex(case["tau"]))
    y = ks.take(1500)
    assert same_bits(y, unhex(case["y"]))


def test_envelope_and_maverage_callers(al):
  """lazy_analysis.py:440-520, 569-616: elementwise Stream stages around a GPU filter; bit-exact."""
  data = load_golden("callers.json")
  x = unhex(data["x"])
  for c in data["cases"]:
    if c["fn"] == "envelope":
      y = list(getattr(al.envelope, c["strategy"])(x, unhex(c["arg"])))
    else:
      filt = getattr(al.maverage, c["strategy"])(c["arg"])
      assert same_bits(filt.numlist, unhex(c["b"])) and same_bits(filt.denlist, unhex(c["a"]))
      y = list(filt(x))
    assert same_bits(y, unhex(c["y"])), (c["fn"], c["strategy"], c["arg"])


def test_vector_valued_samples_like_the_reference(al):
  """tests/golden/multichannel.json was produced with the reference's own multichannel idiom
  (test_filters_extdep.py:49-89): rows as samples, ``zero`` a row, per-channel coefficients as
  ``repeat(ndarray)`` Streams.  The same expression here, bit for bit; and rows through one
  shared LTI filter equal the per-channel scalar runs."""
  mc = load_golden("multichannel.json")
  z, repeat = al.z, al.repeat
  C = mc["C"]
  b = np.array([unhex(r) for r in mc["b"]])
  a = np.array([unhex(r) for r in mc["a"]])
  x = np.array([unhex(r) for r in mc["x"]])
  want = np.array([unhex(r) for r in mc["y"]])
  num = sum(repeat(b[:, k].copy()) * z ** -k for k in range(3) if np.any(b[:, k] != 0))
  den = 1 + sum(repeat(a[:, k].copy()) * z ** -k for k in (1, 2))
  got = np.array(list((num / den)(iter(x), zero=np.zeros(C))))
  assert same_bits(got, want)
  s, Hz = al.sHz(48000)
  filt = al.resonator.z_exp(1000 * Hz, 100 * Hz)
  rows = np.array(list(filt(iter(x), zero=np.full(C, .125), memory=[np.arange(C) * .01, .5])))
  for c in range(C):
    assert same_bits(rows[:, c], list(filt(list(x[:, c]), zero=.125, memory=[c * .01, .5]))), c


def test_calls_own_their_state_and_touch_their_input_lazily(al):
  """Every filter call owns its state like every reference call owns its generator's locals, and
  the input is not pulled before the result is iterated (reference lazy_filters.py:251-262)."""
  from oracle import oracle
  rng = np.random.default_rng(31)
  xa, xb = rng.uniform(-1, 1, 300).tolist(), rng.uniform(-1, 1, 300).tolist()
  b, a = [.2, .3], [1., -.5, .25]
  ref_a, ref_b = oracle.df1(b, a, xa), oracle.df1(b, a, xb)
  bank = al.FilterBank([(b, a)], n_inputs=1)
  ya = bank(xa, block=64)
  first = ya.take(10)                       # the first stream is alive (a block in, 10 items out) ...
  yb = bank(xb, block=64)                   # ... when the same bank is called again
  got_b = list(yb)
  got_a = first + list(ya)
  assert same_bits(got_a, ref_a) and same_bits(got_b, ref_b)
  # laziness: nothing is pulled at call time, one block at the first next()
  pulled = []

  def source():
    for v in xa:
      pulled.append(v)
      yield v
  filt = al.ZFilter(b, a)
  out = filt(source())
  assert pulled == []
  it = iter(out)
  next(it)
  assert 0 < len(pulled) <= al.block_size()
  # a cascade passes rows (vector-valued samples) on like a single filter does
  rows = rng.uniform(-1, 1, (50, 3))
  casc = al.CascadeFilter(al.ZFilter(b, a), al.ZFilter([1., -1.], [1.]))
  got = np.array(list(casc(iter(rows), zero=np.zeros(3))))
  ref = oracle.bank([2, 2], [3, 1], np.array(b + [1., -1.]), np.array(a + [1.]), rows)
  assert same_bits(got, ref)


def test_memory_iterables_on_the_engine(al):
  """Float memories given as one-shot iterators / Streams to INTEGER-coefficient filters: the gate must not draw
  from them (round-3 advisor); read once at call time, member by member in containers (reference
  lazy_filters.py:185-195, :988-990, :1052-1054).  Golden values from the reference (generic_items.json)."""
  gold = {c["tag"]: c for c in load_golden("generic_items.json")}
  acc, two = al.ZFilter([1, 1], [1, -1]), al.ZFilter([1], [1, 0, -1])
  data = [1., 5., -4., -7., 9.]

  def check(tag, res):
    got = list(res)
    assert all(isinstance(v, float) for v in got), tag
    assert [repr(v) for v in got] == gold[tag]["reprs"], tag
  check("mem_iter_float", acc(data, memory=iter([5.0])))
  check("mem_stream_float", two(data, memory=al.Stream([.5, .25, 8.])))
  check("mem_comb_iter_float", (1 / (1 - al.z ** -3))(data + [2.], memory=iter([.5, .25])))
  check("mem_cascade_iter_float", al.CascadeFilter(acc, two, acc)(data, memory=iter([3., 4., 5., 6., 7., 8., 9.])))
  check("mem_parallel_iter_float", al.ParallelFilter(acc, two, acc)(data, memory=iter([3., 4., 5., 6., 7., 8., 9.])))
  mem = [3.]
  res = acc(data[:3], memory=mem)
  mem[0] = 100.            # the call has already staged its memory
  check("mem_mutated_after_call", res)
  # a bad memory argument fails at the call, not at the first next()
  with pytest.raises(TypeError):
    acc(data, memory=5.0)
