"""GPU parity of the elementwise stages: alz_map_dev and the input map fused
into the bank kernels, against vectors produced by the reference (tests/golden/maps.json,
callers.json) and against the oracle on larger seeded blocks.  Bar: bit-exact, except the opt-in
x * x square (one ulp of the reference's libm power, stated in the test)."""
import numpy as np
import pytest

from conftest import load_golden, unhex

pytestmark = pytest.mark.gpu

G = load_golden("maps.json")
X, Y, XS = unhex(G["x"]), unhex(G["y"]), unhex(G["xs"])


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
  return a.shape == b.shape and bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


def test_golden_ops_bit_exact(alz):
  m = alz.maps
  for case in G["unary"]:
    x = unhex(case["x"]) if "x" in case else X
    if case["op"] == "square_pow":
      got, ref = m.square_block(np.array(x)), np.array(unhex(case["r"]))
      assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)) <= 2.3e-16   # x * x vs libm pow: <= 1 ulp
      continue
    if case["op"] == "abs":
      fin = [v for v in x if v == v]           # (the sign of a NaN is not a value)
      assert same_bits(m.abs_block(np.array(fin)), np.abs(np.array(fin)))
    assert same_bits(m.map_block(case["op"], np.array(x)), unhex(case["r"])), case["op"]
  for case in G["scalar"]:
    src = Y if case["op"] == "rdiv" else X
    op = "mul" if case["op"] == "rmul" else case["op"]
    assert same_bits(m.map_block(op, np.array(src), p0=unhex(case["c"])), unhex(case["r"])), case["op"]
  for case in G["binary"]:
    assert same_bits(m.map_block(case["op"] + "2", np.array(X), other=np.array(Y)), unhex(case["r"])), case["op"]
  for case in G["clip"]:
    low = None if case["low"] is None else unhex(case["low"])
    high = None if case["high"] is None else unhex(case["high"])
    assert same_bits(m.clip_block(np.array(X), low, high), unhex(case["r"])), (low, high)


def test_errors_like_the_python_expressions(alz):
  m = alz.maps
  with pytest.raises(ZeroDivisionError):
    m.map_block("div", np.ones(8), p0=0.)
  with pytest.raises(ZeroDivisionError):
    m.map_block("rdiv", np.array([1., 0., 2.]), p0=3.)
  with pytest.raises(ZeroDivisionError):
    m.div_blocks(np.ones(5), np.array([1., 2., 0., 4., 5.]))
  with pytest.raises(ValueError):
    m.clip_block(np.ones(4), 1., -1.)
  with pytest.raises(NotImplementedError):
    m.sqrt_block(np.array([4., -1.]))


def test_large_blocks_vs_oracle_on_the_device(alz, oracle):
  import torch
  rng = np.random.default_rng(17)
  n = (1 << 20) + 3                      # odd length: the scalar tail piece
  x, y = rng.uniform(-2, 2, n), rng.uniform(.5, 2, n)
  xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
  m = alz.maps
  for op, kw in (("abs", {}), ("neg", {}), ("mul", dict(p0=.37)), ("rsub", dict(p0=1.5)), ("rdiv", dict(p0=2.)),
                 ("clip", dict(p0=-.5, p1=.75)), ("clip_low", dict(p0=0.)), ("clip_high", dict(p1=0.))):
    src, srcd = (y, yd) if op == "rdiv" else (x, xd)
    assert same_bits(m.map_block(op, srcd, **kw).cpu().numpy(), oracle.map_block(op, src, **kw)), op
  for op in ("add2", "sub2", "mul2", "div2"):
    assert same_bits(m.map_block(op, xd, other=yd).cpu().numpy(), oracle.map_block(op, x, y)), op
  assert same_bits(m.sqrt_block(yd).cpu().numpy(), np.sqrt(y))
  # unaligned views take the 8-byte path
  assert same_bits(m.abs_block(xd[1:]).cpu().numpy(), np.abs(x[1:]))
  # in place
  z = xd.clone()
  m.map_block("abs", z, out=z)
  assert same_bits(z.cpu().numpy(), np.abs(x))


def test_envelope_strategies_through_the_stream_protocol(alz):
  # reference-generated: envelope.{rms,abs,squared}(xs, 0.02)  (lazy_analysis.py:440-520)
  for case in G["callers"]:
    if not case["fn"].startswith("envelope."):
      continue
    strat = case["fn"].split(".")[1]
    got = list(getattr(alz.envelope, strat)(XS, unhex(case["cutoff"])))
    assert same_bits(got, unhex(case["r"])), strat
  for case in G["callers"]:
    if case["fn"] == "amdf":
      got = list(alz.amdf(case["lag"], case["size"])(XS))
      assert same_bits(got, unhex(case["r"])), case["lag"]


def test_envelope_square_switch(alz, oracle):
  """``envelope.square = "mul"`` (round 5): an existing Stream pipeline squares as ``x * x`` on the device (input map of
  the lowpass kernel, the root as a device map per block) instead of the reference's per-sample ``x ** 2``.  Against the
  reference-generated vectors: within 1e-15 normalised (one ulp in ~8.5e-4 of the squares; contract
  1e-6); against the ``x * x`` definition itself (the oracle on squared input): bit-identical.  Default "pow" unchanged."""
  assert alz.envelope.square == "pow"
  x = np.array(XS)
  try:
    alz.envelope.square = "mul"
    for case in G["callers"]:
      if case["fn"] not in ("envelope.rms", "envelope.squared"):
        continue
      strat, cutoff = case["fn"].split(".")[1], unhex(case["cutoff"])
      got = np.array(list(getattr(alz.envelope, strat)(XS, cutoff)))
      want = np.array(unhex(case["r"]))
      assert got.shape == want.shape and np.max(np.abs(got - want)) <= 1e-15 * np.max(np.abs(want)), strat
      f = alz.lowpass(cutoff)
      ref = oracle.bank([len(f.numlist)], [len(f.denlist)], np.array(f.numlist), np.array(f.denlist), (x * x)[:, None])[:, 0]
      assert same_bits(got, np.sqrt(ref) if strat == "rms" else ref), strat
    # the abs strategy does not depend on the switch
    case = [c for c in G["callers"] if c["fn"] == "envelope.abs"][0]
    assert same_bits(list(alz.envelope.abs(XS, unhex(case["cutoff"]))), unhex(case["r"]))
    alz.envelope.square = "cube"
    with pytest.raises(ValueError):
      alz.envelope.rms(XS, .02)
  finally:
    alz.envelope.square = "pow"
  case = [c for c in G["callers"] if c["fn"] == "envelope.rms"][0]
  assert same_bits(list(alz.envelope.rms(XS, unhex(case["cutoff"]))), unhex(case["r"]))


@pytest.mark.parametrize("layout", ["time", "chan"])
@pytest.mark.parametrize("C", [4096, 16384, 100])
def test_fused_abs_in_front_of_the_lowpass_bank(alz, oracle, layout, C):
  """envelope.abs for a whole bank: |x| on the kernels' input reads (k_duo / k_wave / k_small)."""
  import torch
  N = 4096 + 64 + 6
  rng = np.random.default_rng(C)
  x = rng.uniform(-1, 1, (N, C) if layout == "time" else (C, N))
  cut = rng.uniform(.002, .3, C)
  filts = [alz.lowpass(float(c)) for c in cut]
  b, a = np.array([f.numlist for f in filts]), np.array([f.denlist for f in filts])
  bank = alz.FilterBank([(b, a)], n_inputs=C).set_input_map("abs")
  bank.reset()
  y = bank.process(torch.from_numpy(x).cuda(), layout=layout).cpu().numpy()
  assert "k_duo" in bank.last_kernel or "k_wave" in bank.last_kernel, bank.last_kernel
  ref = oracle.bank([1], [2], b, a, np.abs(x), layout=layout)
  assert same_bits(y, ref), bank.last_kernel
  # the carried input history is the mapped one: a second block continues the reference's stream
  b2 = np.concatenate([b, b * .5], axis=1)            # b0 + b1 z^-1: reads x[-1] across the block edge
  bank2 = alz.FilterBank([(b2, a)], n_inputs=C).set_input_map("abs")
  bank2.reset()
  xd = torch.from_numpy(x).cuda()
  first = (slice(0, 2048), slice(None)) if layout == "time" else (slice(None), slice(0, 2048))
  rest = (slice(2048, None), slice(None)) if layout == "time" else (slice(None), slice(2048, None))
  y1 = bank2.process(xd[first].contiguous(), layout=layout).cpu().numpy()
  y2 = bank2.process(xd[rest].contiguous(), layout=layout).cpu().numpy()
  ref2 = oracle.bank([2], [2], b2, a, np.abs(x), layout=layout)
  assert same_bits(np.concatenate([y1, y2], axis=0 if layout == "time" else 1), ref2)


def test_input_map_in_front_of_shapes_that_do_not_fuse_it(alz, oracle):
  import torch
  C, N = 128, 5000
  rng = np.random.default_rng(5)
  x = rng.uniform(-1, 1, (N, C))
  xd = torch.from_numpy(x).cuda()
  # a cascade (fused cascade kernels / section by section), a long FIR, the time-parallel mode, "neg"
  b, a = np.array([[.2, .1, .05]] * C), np.array([[1., -.5, .25]] * C)
  casc = alz.FilterBank([(b, a), (b, a)], n_inputs=C).set_input_map("abs")
  casc.reset()
  ref = oracle.bank([3, 3], [3, 3], np.concatenate([b, b], 1), np.concatenate([a, a], 1), np.abs(x))
  assert same_bits(casc.process(xd).cpu().numpy(), ref)
  taps = rng.uniform(-1, 1, 40)
  fir = alz.FilterBank([(taps, np.array([1.]))], n_inputs=C).set_input_map("abs")
  fir.reset()
  assert same_bits(fir.process(xd).cpu().numpy(), oracle.bank([40], [1], taps, np.ones(1), np.abs(x)))
  neg = alz.FilterBank([(b, a)], n_inputs=C).set_input_map("neg")
  neg.reset()
  assert same_bits(neg.process(xd).cpu().numpy(), oracle.bank([3], [3], b, a, -x))
  tp = alz.FilterBank([(b, a)], n_inputs=C).set_input_map("abs").set_time_parallel(512)
  tp.reset()
  got = tp.process(xd).cpu().numpy()
  assert "k_scan" in tp.last_kernel
  ref = oracle.bank([3], [3], b, a, np.abs(x))
  assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) <= 1e-12
  # x is left untouched by all of this
  assert same_bits(xd.cpu().numpy(), x)


def test_envelope_block(alz, oracle):
  import torch
  C, N = 512, 8192
  x = np.random.default_rng(2).uniform(-1, 1, (N, C))
  xd = torch.from_numpy(x).cuda()
  cutoff = .01
  f = alz.lowpass(cutoff)
  b, a = np.array(f.numlist), np.array(f.denlist)
  env = alz.envelope_block(xd, cutoff, strategy="abs").cpu().numpy()
  assert same_bits(env, oracle.bank([1], [2], b, a, np.abs(x)))
  with pytest.raises(ValueError):
    alz.envelope_block(xd, cutoff, strategy="rms")          # x * x is opt-in
  rms = alz.envelope_block(xd, cutoff, strategy="rms", square="mul").cpu().numpy()
  ref = np.sqrt(oracle.bank([1], [2], b, a, x * x))
  assert same_bits(rms, ref)                                 # exact with respect to the x * x definition
  # and against the reference's own per-sample form (x ** 2 is libm pow): within a few ulp
  one = np.array(list(alz.envelope.rms(x[:, 0].tolist(), cutoff)))
  assert np.max(np.abs(rms[:, 0] - one)) / np.max(np.abs(one)) <= 1e-15


def test_root_edges_and_out_validation():
  """(-0.0) ** .5 == +0.0 and (-inf) ** .5 == +inf as in CPython (round-2 advisor); ``out`` is validated like the
  inputs on the torch path and honoured on the empty NumPy path."""
  import torch
  from audiolazy_amd import maps as m
  items = [0.0, -0.0, 4.0, 2.0, float("inf"), float("-inf"), 5e-324]
  want = np.array([v ** .5 for v in items])
  got = m.sqrt_block(np.array(items))
  assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
  xd = torch.from_numpy(np.array(items * 40)).cuda()
  got = m.sqrt_block(xd).cpu().numpy()
  assert np.array_equal(got.view(np.uint64), np.tile(want, 40).view(np.uint64))
  for bad in (torch.empty(xd.shape, dtype=torch.float32, device="cuda"), torch.empty((3,), dtype=torch.float64, device="cuda"),
              torch.empty((2 * xd.numel(),), dtype=torch.float64, device="cuda")[::2], torch.empty(xd.shape, dtype=torch.float64)):
    with pytest.raises(ValueError):
      m.abs_block(xd, out=bad)
  out = torch.empty_like(xd)
  assert m.abs_block(xd, out=out) is out
  empty_out = np.empty((0,))
  assert m.abs_block(np.empty((0,)), out=empty_out) is empty_out
