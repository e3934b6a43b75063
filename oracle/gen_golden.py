#!/usr/bin/env python3
"""Generate tests/golden/*.json by RUNNING THE REFERENCE (AudioLazy 0.6.1dev).

Run in the build container only (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Every vector committed under tests/golden/ comes out of the unmodified
reference imported from /root/reference; this script is the provenance.
Floats are stored as ``float.hex()`` strings so the fixtures are bit-exact.

Reference entry points exercised (paths relative to /root/reference/):
  audiolazy/lazy_filters.py:141-264   LinearFilter.__call__ (DF-I generator)
  audiolazy/lazy_filters.py:988-990   CascadeFilter.__call__
  audiolazy/lazy_filters.py:1048-1054 ParallelFilter.__call__
  audiolazy/lazy_filters.py:1087-1495 comb / resonator / lowpass / highpass
  audiolazy/lazy_auditory.py:55-218   erb, gammatone_erb_constants, gammatone
  audiolazy/lazy_analysis.py:277-312  acorr
  audiolazy/lazy_lpc.py:52-136,229-272 levinson_durbin, lpc.kautocor
  audiolazy/lazy_misc.py:74-129       blocks
  audiolazy/lazy_wav.py:31-130        WavStream
  audiolazy/lazy_io.py:44-94          chunks.struct
"""
import itertools
import json
import os
import random
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import audiolazy as al  # noqa: E402
from audiolazy import (z, ZFilter, CascadeFilter, ParallelFilter, Stream,  # noqa: E402
                       resonator, comb, lowpass, highpass, sHz, gammatone, erb,
                       gammatone_erb_constants, acorr, levinson_durbin, lpc,
                       white_noise, repeat)
import numpy as np  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
s, Hz = sHz(48000)


def hx(v):
  if isinstance(v, (list, tuple)):
    return [hx(i) for i in v]
  return float(v).hex()


def noise(n, seed):
  random.seed(seed)
  return [random.uniform(-1.0, 1.0) for _ in range(n)]


def dump(name, obj):
  path = os.path.join(OUT, name)
  with open(path, "w") as f:
    json.dump(obj, f, indent=0, separators=(",", ":"))
  print("%-28s %8d bytes" % (name, os.path.getsize(path)))


# --------------------------------------------------------------------------
# 1. DF-I execution: single filters, seeded inputs, memory / zero variants
# --------------------------------------------------------------------------
def filt_cases():
  cases = []

  def add(tag, filt, x, **kw):
    y = list(filt(list(x), **kw))
    mem = kw.get("memory")
    if callable(mem):
      raise ValueError
    case = dict(tag=tag, b=hx(filt.numlist), a=hx(filt.denlist),
                memory=None if mem is None else hx(list(mem)),
                zero=hx(kw.get("zero", 0.0)), y=hx(y))
    if list(x) == base[:len(x)]:
      case["x_len"] = len(x)  # prefix of the shared input
    else:
      case["x"] = hx(x)
    cases.append(case)

  x = base = noise(400, 1234)
  add("lowpass.pole**2@1k", lowpass.pole(1000 * Hz) ** 2, x)
  add("lowpass.pole@1k", lowpass.pole(1000 * Hz), x)
  add("lowpass.z@3k", lowpass.z(3000 * Hz), x)
  add("lowpass.pole_exp@500", lowpass.pole_exp(500 * Hz), x)
  add("lowpass.z_exp@500", lowpass.z_exp(500 * Hz), x)
  add("highpass.z@1k", highpass.z(1000 * Hz), x)
  add("highpass.pole@2k", highpass.pole(2000 * Hz), x)
  add("highpass.pole_exp@2k", highpass.pole_exp(2000 * Hz), x)
  add("highpass.z_exp@2k", highpass.z_exp(2000 * Hz), x)
  add("resonator.z_exp@1k/100", resonator.z_exp(1000 * Hz, 100 * Hz), x)
  add("resonator.poles_exp@1k/100", resonator.poles_exp(1000 * Hz, 100 * Hz), x)
  add("resonator.freq_z_exp@440/30", resonator.freq_z_exp(440 * Hz, 30 * Hz), x)
  add("resonator.freq_poles_exp@440/30", resonator.freq_poles_exp(440 * Hz, 30 * Hz), x)
  add("resonator.z_exp@50/1 highQ", resonator.z_exp(50 * Hz, 1 * Hz), x)
  add("comb.fb(5,.5)", comb.fb(5, .5), x)
  add("comb.ff(7,-.25)", comb.ff(7, -.25), x)
  add("comb.tau(20,100)", comb.tau(20, 100), x)
  add("gain a0=2", ZFilter([.2, .3, .4], [2., -.5, .25]), x)
  add("gain a0=-1", ZFilter([.2, .3, .4], [-1., -.5, .25]), x)
  add("gain a0=-18 sparse", ZFilter([1., 0., -1.], [-18., 9.8, 0., 14.3]), x)
  add("fir4", ZFilter([.1, .2, .3, .4]), x)
  add("fir unit taps", ZFilter([1., -1., 0., 1.]), x)
  add("pure delay 3", z ** -3 + 0., x)
  add("b0==0 leading", ZFilter([0., 0.5, 0.25], [1., -0.3]), x)
  add("a1==0", ZFilter([0.5], [1., 0., 0.81]), x)
  add("unit feedback", 1 / (1 - z ** -1), x[:64])
  add("x[n-1]-y[n-2]", z ** -1 / (1 + z ** -2), x[:64])
  # memory / zero semantics (lazy_filters.py:185-195, 243-250)
  f2 = resonator.z_exp(1000 * Hz, 100 * Hz)
  add("mem full", f2, x[:100], memory=[0.25, -0.5])
  add("mem short -> left pad", f2, x[:100], memory=[0.7])
  add("mem long -> truncated", f2, x[:100], memory=[0.1, 0.2, 0.3, 0.4])
  add("zero=0.5", f2, x[:100], zero=0.5)
  add("mem short + zero", f2, x[:100], memory=[0.7], zero=-0.25)
  add("fir zero=1.5", ZFilter([.1, .2, .3, .4]), x[:50], zero=1.5)
  add("gain + mem + zero", ZFilter([.2, .3, .4], [2., -.5, .25]), x[:100],
      memory=[1.0, -1.0], zero=0.125)
  # an 8-tap-numerator section of gammatone.sampled (ill-conditioned numerator)
  g = gammatone.sampled(100 * Hz, gammatone_erb_constants(4)[0] * erb(100 * Hz, Hz))
  add("gammatone.sampled[0]@100", g[0], x)
  # non-finite input: zero-coefficient terms are absent from the expression
  xin = list(x[:40]); xin[7] = float("inf")
  add("inf with zero tap", ZFilter([0.5, 0.0, 0.25]), xin)
  return dict(x=hx(base), cases=cases)


# --------------------------------------------------------------------------
# 2. scipy.signal.lfilter grid re-stated from tests/test_filters_extdep.py:41-47
#    (that file cannot be collected on NumPy 2; run its cases on the reference)
# --------------------------------------------------------------------------
def lfilter_grid():
  out = []
  for a in ([1.], [3.], [1., 3.], [15., -17.2], [-18., 9.8, 0., 14.3]):
    for b in ([1.], [-1.], [1., 0., -1.], [1., 3.]):
      for data in (list(range(5)), list(range(5, 0, -1)), [7, 22, -5], [8., 3., 15.]):
        y = list(ZFilter(b, a)([float(v) for v in data]))
        out.append(dict(b=hx(b), a=hx(a), x=hx(data), y=hx(y)))
  return out


# --------------------------------------------------------------------------
# 3. containers
# --------------------------------------------------------------------------
def container_cases():
  x = noise(400, 99)
  out = []
  fs = [lowpass.pole(800 * Hz), highpass.z(200 * Hz), resonator.z_exp(1500 * Hz, 80 * Hz)]
  casc = CascadeFilter(fs)
  out.append(dict(kind="cascade", sections=[dict(b=hx(f.numlist), a=hx(f.denlist)) for f in fs],
                  x=hx(x), memory=None, zero=hx(0.0), y=hx(list(casc(list(x))))))
  # same memory / zero forwarded to every stage (lazy_filters.py:989)
  fs2 = [resonator.poles_exp(700 * Hz, 50 * Hz), resonator.z_exp(900 * Hz, 60 * Hz)]
  out.append(dict(kind="cascade", sections=[dict(b=hx(f.numlist), a=hx(f.denlist)) for f in fs2],
                  x=hx(x[:120]), memory=hx([0.3, -0.2]), zero=hx(0.1),
                  y=hx(list(CascadeFilter(fs2)(list(x[:120]), memory=[0.3, -0.2], zero=0.1)))))
  par = ParallelFilter(fs)
  out.append(dict(kind="parallel", sections=[dict(b=hx(f.numlist), a=hx(f.denlist)) for f in fs],
                  x=hx(x), memory=None, zero=hx(0.0), y=hx(list(par(list(x))))))
  return out


# --------------------------------------------------------------------------
# 4. design functions: coefficients only
# --------------------------------------------------------------------------
def design_cases():
  out = []
  freqs = [50., 440., 1000., 3000., 12000., 20000.]
  for name, sd in (("lowpass", lowpass), ("highpass", highpass)):
    for strat in ("pole", "z", "pole_exp", "z_exp"):
      for f in freqs:
        filt = getattr(sd, strat)(f * Hz)
        out.append(dict(fn=name, strategy=strat, args=hx([f * Hz]),
                        b=hx(filt.numlist), a=hx(filt.denlist)))
  for strat in ("poles_exp", "freq_poles_exp", "z_exp", "freq_z_exp"):
    for f in freqs:
      for q in (2., 10., 50.):
        filt = getattr(resonator, strat)(f * Hz, f / q * Hz)
        out.append(dict(fn="resonator", strategy=strat, args=hx([f * Hz, f / q * Hz]),
                        b=hx(filt.numlist), a=hx(filt.denlist)))
  for d, al_ in ((1, .5), (5, .5), (100, -.9), (441, .99)):
    for strat in ("fb", "ff"):
      filt = getattr(comb, strat)(d, al_)
      out.append(dict(fn="comb", strategy=strat, args=hx([d, al_]),
                      b=hx(filt.numlist), a=hx(filt.denlist)))
  for d, tau in ((20, 100), (109, 48000)):
    filt = comb.tau(d, tau)
    out.append(dict(fn="comb", strategy="tau", args=hx([d, tau]),
                    b=hx(filt.numlist), a=hx(filt.denlist)))
  return out


def auditory_cases():
  out = dict(erb=[], consts=[], gammatone=[])
  for strat in ("gm90", "mg83"):
    for f in (50., 100., 1000., 3000., 8000., 20000.):
      out["erb"].append(dict(strategy=strat, freq=hx(f), value=hx(getattr(erb, strat)(f))))
      out["erb"].append(dict(strategy=strat, freq=hx(f * Hz), Hz=hx(Hz),
                             value=hx(getattr(erb, strat)(f * Hz, Hz))))
  for n in (1, 2, 3, 4, 5, 8):
    out["consts"].append(dict(n=n, value=hx(gammatone_erb_constants(n))))
  x = noise(500, 777)
  for strat in ("sampled", "slaney", "klapuri"):
    for fc in (50., 100., 1000., 8000., 20000.):
      bw = gammatone_erb_constants(4)[0] * erb(fc * Hz, Hz)
      g = getattr(gammatone, strat)(fc * Hz, bw)
      y = list(g(list(x)))
      out["gammatone"].append(dict(strategy=strat, freq=hx(fc * Hz), bw=hx(bw),
                                   sections=[dict(b=hx(f.numlist), a=hx(f.denlist)) for f in g],
                                   y=hx(y)))
  out["x"] = hx(x)
  return out


# --------------------------------------------------------------------------
# 5. multichannel: the reference's vector-valued-sample idiom
#    (tests/test_filters_extdep.py:49-89; per-channel coefs via repeat(ndarray))
# --------------------------------------------------------------------------
def multichannel_case():
  rng = np.random.default_rng(20260924)
  C, N = 8, 256
  x = rng.uniform(-1, 1, (N, C))
  fcs = np.geomspace(50., 20000., C)
  filts = [resonator.z_exp(fc * Hz, fc / 10 * Hz) for fc in fcs]
  b = np.array([f.numlist for f in filts])
  a = np.array([f.denlist for f in filts])
  num = sum(repeat(b[:, k].copy()) * z ** -k for k in range(3) if np.any(b[:, k] != 0))
  den = 1 + sum(repeat(a[:, k].copy()) * z ** -k for k in (1, 2))
  filt = num / den
  y = np.array(list(filt(iter(x), zero=np.zeros(C))))
  # each channel also through the scalar path: must be identical
  for c in range(C):
    ys = list(filts[c](list(x[:, c])))
    assert ys == list(y[:, c]), "vector-valued idiom differs from scalar path"
  return dict(C=C, N=N, b=[hx(r) for r in b.tolist()], a=[hx(r) for r in a.tolist()],
              x=[hx(r) for r in x.tolist()], y=[hx(r) for r in y.tolist()])


# --------------------------------------------------------------------------
# 6. LPC
# --------------------------------------------------------------------------
def lpc_cases():
  out = dict(acorr=[], levinson=[], kautocor=[])
  for seq, lag in (([1, 2, 3, 4, 3, 4, 2], None), ([1, 2, 3, 4, 3, 4, 2], 9),
                   ([2, 2, 0, 0, -1, -1, 0, 0, 1, 1], None)):
    out["acorr"].append(dict(x=hx(seq), max_lag=lag, r=hx(acorr(seq, lag) if lag else acorr(seq))))
  fr = noise(480, 4242)
  out["acorr"].append(dict(x=hx(fr), max_lag=16, r=hx(acorr(fr, 16))))
  for ac, order in (([12, 6, 0, -3, -6, -3, 0, 2, 4, 2], 3), ([1, 5, 3], None),
                    ([1., .5, .25, .125], 5), (acorr(fr, 16), 16)):
    f = levinson_durbin(ac, order) if order else levinson_durbin(ac)
    out["levinson"].append(dict(ac=hx(ac), order=order, coefs=hx(f.numlist), error=hx(f.error)))
  random.seed(31337)
  res = resonator.z_exp(700 * Hz, 40 * Hz)
  ar = list(res(noise(480 + 200, 5))) [200:]
  frames = [fr, ar, [-1., 0., 1., 0.] * 4, noise(480, 6), [float(i % 7) - 3. for i in range(480)]]
  for blk in frames:
    for order in ((2, 16) if len(blk) > 16 else (2,)):
      f = lpc.kautocor(blk, order)
      out["kautocor"].append(dict(x=hx(blk), order=order, coefs=hx(f.numlist), error=hx(f.error)))
  return out


def lpc_strategy_cases():
  """The other autocorrelation-method strategies (lazy_lpc.py:140-225): ``lpc(blk, order)`` = ``lpc.autocor``
  (pseudo-inverse form below order 100, Levinson-Durbin from 100 up with the pseudo-inverse as ParCorError
  fallback), ``lpc.nautocor``; ``lpc.kautocor`` and ``acorr`` past 63 lags.  Blocks are stored once."""
  fr = noise(480, 4242)
  res = resonator.z_exp(700 * Hz, 40 * Hz)
  ar = list(res(noise(480 + 200, 5)))[200:]
  silent = [0.] * 150          # kautocor raises ParCorError -> lpc.autocor falls back to the pseudo-inverse form
  blocks = dict(noise=fr, resonant=ar, silent=silent)
  out = dict(blocks={k: hx(v) for k, v in blocks.items()}, acorr=[], nautocor=[], autocor=[], kautocor=[])
  for lag in (64, 80, 130):
    out["acorr"].append(dict(blk="noise", max_lag=lag, r=hx(acorr(fr, lag))))
  for name in ("noise", "resonant"):
    blk = blocks[name]
    for order in (2, 8, 16):
      f = lpc.nautocor(blk, order)
      out["nautocor"].append(dict(blk=name, order=order, coefs=hx(f.numlist), error=hx(f.error)))
    for order in (16, 70, 100, 120):
      f = lpc(blk, order)
      out["autocor"].append(dict(blk=name, order=order, coefs=hx(f.numlist), error=hx(f.error),
                                 route="nautocor" if order < 100 else "kautocor"))
    for order in (40, 70, 100):
      f = lpc.kautocor(blk, order)
      out["kautocor"].append(dict(blk=name, order=order, coefs=hx(f.numlist), error=hx(f.error)))
  f = lpc(silent, 100)
  out["autocor"].append(dict(blk="silent", order=100, coefs=hx(f.numlist), error=hx(f.error), route="fallback"))
  return out


def covariance_cases():
  """The covariance-method callers beside the path: ``lag_matrix`` (lazy_analysis.py:315-342), ``lpc.covar`` /
  ``lpc.kcovar`` (lazy_lpc.py:275-340), ``parcor`` / ``parcor_stable`` (:343-425), ``toeplitz`` (:44-49).  Float
  results as hex, integer ones as ``repr``; every case that raises stores the exception's type and text.  ``lsf`` /
  ``lsf_stable`` are absent: the reference's own call raises under NumPy 2 (its elementwise ``phase`` on an array)."""
  from audiolazy import lag_matrix, parcor, parcor_stable, toeplitz

  def outcome(fn):
    try:
      return dict(value=fn())
    except Exception as exc:
      return dict(raises=type(exc).__name__, text=str(exc))

  fr = noise(480, 4242)
  res = resonator.z_exp(700 * Hz, 40 * Hz)
  ar = list(res(noise(480 + 200, 5)))[200:]
  periodic = [-1., 0., 1., 0.] * 50                       # README.rst:360-365: lpc.covar(blk, 4) = 1 + .5 z^-2 - .5 z^-4
  ints = [1, 2, 3, 4, 5, -1, 2, -6, 3, 1]
  mixed = [1, 2.5, -3, 4, .125, -1, 2, -6.75, 3, 1]
  blocks = dict(noise=fr, resonant=ar, periodic=periodic, silent=[0.] * 40, short=noise(9, 11), ints=ints, mixed=mixed,
                ramp=[float(i) for i in range(1, 13)])
  out = dict(blocks={k: (hx(v) if k not in ("ints", "mixed") else repr(v)) for k, v in blocks.items()},
             lag_matrix=[], covar=[], kcovar=[], parcor=[], parcor_stable=[], toeplitz=[])
  for name, lag in (("noise", 2), ("noise", 16), ("resonant", 8), ("short", None), ("short", 0), ("short", 8),
                    ("periodic", 4), ("silent", 3), ("mixed", 3)):
    out["lag_matrix"].append(dict(blk=name, max_lag=lag, phi=hx(lag_matrix(blocks[name], lag))))
  out["lag_matrix"].append(dict(blk="ints", max_lag=2, phi_repr=repr(lag_matrix(ints, 2))))
  out["lag_matrix"].append(dict(blk="ints", max_lag=None, phi_repr=repr(lag_matrix(ints))))
  for lag in (9, 10, 50):
    out["lag_matrix"].append(dict(blk="short", max_lag=lag, **outcome(lambda: hx(lag_matrix(blocks["short"], lag)))))

  def filt_outcome(fn):
    def run():
      f = fn()
      return dict(coefs=hx(f.numlist), first_is_int=isinstance(f.numlist[0], int), den=hx(f.denlist),
                  error=hx(f.error), error_type=type(f.error).__name__)
    return outcome(run)

  for name in ("noise", "resonant", "periodic", "ramp", "ints", "mixed", "silent", "short"):
    for order in ((1, 2, 4, 8, 16) if name in ("noise", "resonant") else (1, 2, 4)):
      out["covar"].append(dict(blk=name, order=order, **filt_outcome(lambda: lpc.covar(blocks[name], order))))
      out["kcovar"].append(dict(blk=name, order=order, **filt_outcome(lambda: lpc.kcovar(blocks[name], order))))
  for alias in ("cov", "covariance", "ncovar", "ncov", "ncovariance", "kcov", "kcovariance"):
    f = lpc[alias](fr, 3)
    out["covar"].append(dict(blk="noise", order=3, alias=alias, value=dict(
        coefs=hx(f.numlist), first_is_int=isinstance(f.numlist[0], int), den=hx(f.denlist), error=hx(f.error),
        error_type=type(f.error).__name__)))

  def parcor_outcome(filt):
    got = []
    try:
      for k in parcor(filt):
        got.append(k)
      return dict(ks=hx(got))
    except Exception as exc:
      return dict(ks=hx(got), raises=type(exc).__name__, text=str(exc))

  designs = dict(
    doctest=lambda: levinson_durbin([1, 2, 3, 4, 5, 3, 2, 1]),
    kautocor8=lambda: lpc.kautocor(fr, 8),
    kautocor16=lambda: lpc.kautocor(ar, 16),
    gain2=lambda: ZFilter([2., 1., .5, -.25], [2.]),
    unit_tap=lambda: 1 + z ** -1,                          # k = 1: 1 - k**2 = 0 -> ParCorError after the first value
    mid_unit=lambda: 1 + .5 * z ** -1 + 1. * z ** -2,
    feedback=lambda: (1 + .5 * z ** -1) / (1 - .3 * z ** -1),
    constant=lambda: ZFilter([1.]),
    ints=lambda: ZFilter([1, 2, 3]),
  )
  for name, make in designs.items():
    f = make()
    out["parcor"].append(dict(name=name, num=hx(f.numlist), den=hx(f.denlist), **parcor_outcome(f)))
  stable_designs = dict(
    resonator=lambda: resonator.z_exp(700 * Hz, 40 * Hz),
    lowpass=lambda: lowpass.pole(1000 * Hz),
    kautocor_inverse=lambda: 1 / lpc.kautocor(ar, 12),
    outside=lambda: 1 / (1 - 2.5 * z ** -1 + z ** -2),
    critical=lambda: 1 / (1 - z ** -1),
    oscillator=lambda: 1 / (1 - 1.2 * z ** -1 + z ** -2),
    fir=lambda: 1 + .5 * z ** -1,
    double_pole=lambda: 1 / (1 - .9 * z ** -1) ** 2,
  )
  for name, make in stable_designs.items():
    f = make()
    out["parcor_stable"].append(dict(name=name, num=hx(f.numlist), den=hx(f.denlist), **outcome(lambda: parcor_stable(f))))
  for vect in ([1, 2, 3], [1.5], [], [4., -2., 0., 1.]):
    out["toeplitz"].append(dict(vect=repr(vect), matrix=repr(toeplitz(vect))))
  return out


def lagrange_cases():
  """``lagrange.func`` / ``lagrange.poly`` / ``resample`` (lazy_poly.py:493-603) on seeded point sets and signals: values as
  ``repr`` (ints stay ints, Fractions are not used), Poly terms as sorted (power, value) reprs, resampled items as hex."""
  from audiolazy import lagrange, resample
  rnd = random.Random(20260927)
  out = dict(func=[], poly=[], resample=[])
  for case in range(120):
    n = rnd.randint(1, 6)
    xs = rnd.sample(range(-8, 9), n) if case % 2 else [rnd.uniform(-3, 3) for _ in range(n)]
    ys = [rnd.choice([rnd.randint(-5, 5), rnd.uniform(-2, 2)]) for _ in range(n)]
    pairs = [list(pr) for pr in zip(xs, ys)]
    ks = [0, 1.5, -2, 7, rnd.uniform(-4, 4)]

    def value(k):
      try:
        return repr(lagrange([tuple(pr) for pr in pairs])(k))
      except Exception as exc:
        return "raises " + type(exc).__name__
    out["func"].append(dict(pairs=repr(pairs), ks=repr(ks), values=[value(k) for k in ks]))
    try:
      terms = repr(sorted(lagrange.poly([tuple(pr) for pr in pairs]).terms()))
    except Exception as exc:
      terms = "raises " + type(exc).__name__
    out["poly"].append(dict(pairs=repr(pairs), terms=terms))
  for case in range(40):
    sig = [rnd.uniform(-1, 1) for _ in range(60)]
    old, new = rnd.choice([(1, 1), (1, 2), (3, 2), (1, 1.7), (2, 1), (1, .5), (44100, 48000)])
    order = rnd.choice([1, 2, 3, 4, 5])
    zero = rnd.choice([0., 0, .25])
    got = resample(list(sig), old=old, new=new, order=order, zero=zero).take(12)
    out["resample"].append(dict(sig=hx(sig), old=old, new=new, order=order, zero=repr(zero), y=hx(got)))
  sig = [rnd.uniform(-1, 1) for _ in range(80)]
  steps = [rnd.choice([.5, 1., 1.25, .75]) for _ in range(30)]
  got = resample(list(sig), old=Stream(list(steps)), new=1, order=3).take(20)      # a time-varying step, one value per output
  out["resample"].append(dict(sig=hx(sig), old_series=hx(steps), new=1, order=3, zero=repr(0.), y=hx(got)))
  return out


def generic_item_cases():
  """Items the float64 engine does not take, through the reference's type-generic generator
  (lazy_filters.py:141-264): all-integer calls keep ints (doctest :735-742), complex numbers, Fractions,
  NumPy matrices with matrix-valued coefficient Streams (tests/test_filters_extdep.py:49-89).  Results are
  stored as ``repr`` strings: exact for every type involved."""
  from fractions import Fraction
  from math import pi, cos, sqrt
  from audiolazy import cycle, count
  out = []

  def add(tag, result):
    items = list(result)
    out.append(dict(tag=tag, types=[type(v).__name__ for v in items],
                    reprs=[repr(v.tolist()) if hasattr(v, "tolist") else repr(v) for v in items]))
  filt = ZFilter([1, 1], [1, -1])
  add("int_doctest", filt([1, 5, -4, -7, 9], memory=[3], zero=0))
  add("int_doctest_delayed", (filt * z ** -1)([4, 10, 11, 0, 2], zero=0))
  add("int_default_zero_is_float", filt([1, 5, -4, -7, 9]))
  add("int_gain_only", ZFilter([3])([1, 2, -5]))
  add("int_then_float_items", ZFilter([3, 1])([1, 2.5, 2, 7], zero=0))
  add("int_division_by_gain", ZFilter([1, 1], [2, -1])([1, 5, -4, -7, 9], zero=0))
  add("int_negative_gain", ZFilter([2, 1], [-1, 3])([1, 5, -4], zero=0))
  add("complex_items", (1 - z ** -1)([1j, 2, 3 + 1j, -1.5j]))
  add("complex_coefficients", ZFilter([1, .5j], [1, -.25 + .1j])([1j, 2, 3 + 1j, -1.5j, 0, 1]))
  add("fraction_items", ZFilter([1, 2], [1, -1])([Fraction(1, 3), Fraction(2, 7), Fraction(-5, 2)], zero=0))
  add("complex_series", (Stream(itertools.cycle([1j, 2.])) + z ** -1)([1., 2., 3., 4.]))
  # the matrix case of tests/test_filters_extdep.py:49-89 (numpy.matrix: numpy.mat is gone from NumPy 2)
  mat = np.matrix
  m, n1, n2 = mat([[1, 2], [2, 2]]), mat([[1.2, 3.2], [1.2, 1.1]]), mat([[-1, 2], [-1, 2]])
  a = mat([[.3, .4], [.5, .6]])
  filt = (repeat(m) + cycle([n1, n2]) * z ** -1) / (1 - repeat(a) * z ** -1)
  data = [Stream(1, 2), count(), count(start=1, step=2), cycle([.2, .33, .77, pi, cos(3)]), repeat(pi),
          count(start=sqrt(2), step=pi / 3)]
  sig = Stream(mat(vect).reshape(2, 3) for vect in zip(*data))
  add("matrix_items_matrix_series", filt(sig, zero=mat([[0, 0, 0], [0, 0, 0]])).limit(12))
  # ``memory`` given as a one-shot iterator / Stream / callable: read once, at call time (:185-195); containers
  # hand the SAME object to every member in turn (:988-990, :1052-1054), so an iterator is drawn from member by
  # member (the ``takewhile`` that reads it also draws the item that ends it)
  acc = ZFilter([1, 1], [1, -1])
  two = ZFilter([1], [1, 0, -1])
  add("mem_iter_int", acc([1, 5, -4, -7, 9], memory=iter([3]), zero=0))
  add("mem_stream_int", acc([1, 5, -4, -7, 9], memory=Stream([3, 8]), zero=0))
  add("mem_iter_short_int", two([1, 5, -4, -7, 9], memory=iter([3]), zero=0))
  add("mem_callable_int", two([1, 5, -4, -7, 9], memory=lambda n: [7] * n, zero=0))
  add("mem_cascade_iter_int", CascadeFilter(acc, two, acc)([1, 5, -4, -7, 9], memory=iter([3, 4, 5, 6, 7, 8, 9]), zero=0))
  add("mem_cascade_list_int", CascadeFilter(acc, two, acc)([1, 5, -4, -7, 9], memory=[3, 4, 5], zero=0))
  add("mem_parallel_iter_int", ParallelFilter(acc, two, acc)([1, 5, -4, -7, 9], memory=iter([3, 4, 5, 6, 7, 8, 9]), zero=0))
  # float memories on integer-coefficient filters: these calls are the float engine's (GPU tests read them)
  add("mem_iter_float", acc([1., 5., -4., -7., 9.], memory=iter([5.0])))
  add("mem_stream_float", two([1., 5., -4., -7., 9.], memory=Stream([.5, .25, 8.])))
  add("mem_comb_iter_float", (1 / (1 - z ** -3))([1., 5., -4., -7., 9., 2.], memory=iter([.5, .25])))
  add("mem_cascade_iter_float", CascadeFilter(acc, two, acc)([1., 5., -4., -7., 9.], memory=iter([3., 4., 5., 6., 7., 8., 9.])))
  add("mem_parallel_iter_float", ParallelFilter(acc, two, acc)([1., 5., -4., -7., 9.], memory=iter([3., 4., 5., 6., 7., 8., 9.])))
  add("mem_mutated_after_call", _mutated_memory_case())
  # constants travel through the generated SOURCE as text (lazy_filters.py:209, 224, 229-231, 236): what is not its own
  # literal is re-read by the parser -- Fraction(3, 5) -> ``3/5`` (float arithmetic), a Fraction gain a/b -> ``(...) / a/b``,
  # -1j -> ``(-0-1j)`` (positive zero real part), numpy / Decimal scalars -> plain floats, ``zero`` of a term-less filter
  # -> the literal of its text (round 5)
  from decimal import Decimal
  add("text_fraction_coefficients", ZFilter([Fraction(3, 5), 1], [1, Fraction(-1, 4)])([1, 2, 3], zero=0))
  add("text_fraction_gain", ZFilter([1, 2], [Fraction(3, 2), -1])([1, 2, 3], zero=0))
  add("text_fraction_coefficient_fraction_items", ZFilter([Fraction(3, 5)], [1])([Fraction(1, 3), Fraction(2, 7)], zero=0))
  add("text_negative_fraction_denominator", ZFilter([1], [1, Fraction(-3, 7), Fraction(2, 9)])([1, 2, 3, 4], zero=0))
  add("text_zero_fraction_whole", ZFilter([0], [1])([1, 2], zero=Fraction(3)))
  add("text_zero_fraction", ZFilter([0], [1])([1, 2], zero=Fraction(-5, 4)))
  add("text_zero_complex", ZFilter([0], [1])([1, 2], zero=-2j))
  add("text_gain_minus_1j", ZFilter([1, 1], [-1j, .5])([1., 2., 3.]))
  add("text_gain_2j", ZFilter([1, 1], [2j, .5])([1., 2., 3.]))
  add("text_complex_denominator", ZFilter([1], [1, -2j])([1., 2., 3.]))
  add("text_bool_coefficient", ZFilter([True, 2], [1])([1, 2, 3], zero=0))
  add("text_numpy_int_coefficient", ZFilter([np.int64(3), 2], [1])([1, 2, 3], zero=0))
  add("text_numpy_complex_coefficient", ZFilter([np.complex128(1 + 2j), 2], [1])([1, 2, 3], zero=0))
  add("text_numpy_float_coefficient_complex_items", ZFilter([np.float64(.1), 2], [1, np.float64(-.5)])([1j, 2, 3 - 1j]))
  add("text_decimal_coefficient", ZFilter([Decimal("0.1"), 2], [1])([1, 2, 3], zero=0))
  return out


def _mutated_memory_case():
  """The memory list is changed between the call and the iteration: the reference has already copied it."""
  mem = [3.]
  res = ZFilter([1, 1], [1, -1])([1., 5., -4.], memory=mem)
  mem[0] = 100.
  return res


# --------------------------------------------------------------------------
# 7. Stream.blocks
# --------------------------------------------------------------------------
def blocks_cases():
  out = []
  for n, size, hop, pad in ((10, 4, None, 0.), (10, 4, 2, 0.), (10, 4, 6, 0.), (7, 3, 3, -1.),
                            (12, 4, 4, 0.), (5, 8, None, 9.), (9, 4, 1, 0.), (11, 5, 7, 0.5)):
    data = [float(i) for i in range(n)]
    kw = dict(size=size, padval=pad)
    if hop is not None:
      kw["hop"] = hop
    blks = [list(b) for b in Stream(data).blocks(**kw)]
    out.append(dict(n=n, size=size, hop=hop, padval=pad, blocks=blks))
  return out


# --------------------------------------------------------------------------
# 8. karplus_strong (lazy_synth.py:624-657): comb.tau at a fractional period, linearize(),
#    callable memory (white_noise) -- seeded so the noise is reproducible
# --------------------------------------------------------------------------
def karplus_case():
  from audiolazy import karplus_strong
  out = []
  for seed, f, tau in ((77, 440., 2e4), (5, 1234.5, 3000.)):
    random.seed(seed)
    ks = karplus_strong(f * Hz, tau)
    y = ks.take(1500)
    filt = comb.tau(2 * 3.141592653589793 / (f * Hz), tau).linearize()
    out.append(dict(seed=seed, freq=hx(f * Hz), tau=hx(tau), b=hx(filt.numlist), a=hx(filt.denlist), y=hx(y)))
  return out


# --------------------------------------------------------------------------
# 9. callers of the filter path: envelope (lazy_analysis.py:440-520), maverage (:569-616)
# --------------------------------------------------------------------------
def callers_case():
  from audiolazy import envelope, maverage
  x = noise(400, 2024)
  out = dict(x=hx(x), cases=[])
  for strat in ("rms", "abs", "squared"):
    out["cases"].append(dict(fn="envelope", strategy=strat, arg=hx(0.05),
                             y=hx(list(getattr(envelope, strat)(list(x), 0.05)))))
  for strat in ("recursive", "fir"):
    for size in (4, 25):
      filt = getattr(maverage, strat)(size)
      out["cases"].append(dict(fn="maverage", strategy=strat, arg=size, b=hx(filt.numlist),
                               a=hx(filt.denlist), y=hx(list(filt(list(x))))))
  return out


# --------------------------------------------------------------------------
# 11. formats either side of the path: WavStream, chunks; ParallelFilter memory rules
# --------------------------------------------------------------------------
def formats_cases():
  import io
  import tempfile
  import wave
  from audiolazy import WavStream, chunks
  out = dict(wav=[], chunks=[], parallel=[])
  rnd = random.Random(77)
  for bits in (8, 16, 24, 32):
    for channels in (1, 2):
      frames = 37
      raw = bytes(rnd.randrange(256) for _ in range(frames * channels * bits // 8))
      # corner values first: most negative, most positive, zero, minus one
      width = bits // 8
      if bits == 8:
        corners = [bytes([0]), bytes([255]), bytes([128]), bytes([127])]
      else:
        corners = [(-(1 << (bits - 1))).to_bytes(width, "little", signed=True),
                   ((1 << (bits - 1)) - 1).to_bytes(width, "little", signed=True),
                   (0).to_bytes(width, "little", signed=True), (-1).to_bytes(width, "little", signed=True)]
      raw = b"".join(corners) + raw[4 * width:]
      with tempfile.NamedTemporaryFile(suffix=".wav") as tmp:
        w = wave.open(tmp.name, "wb")
        w.setnchannels(channels)
        w.setsampwidth(width)
        w.setframerate(22050)
        w.writeframes(raw)
        w.close()
        file_bytes = open(tmp.name, "rb").read()
        ws = WavStream(tmp.name)
        meta = dict(rate=ws.rate, channels=ws.channels, bits=ws.bits)
        scaled = list(ws)
        kept = list(WavStream(tmp.name, keep=True))
      out["wav"].append(dict(meta, file=file_bytes.hex(), raw=raw.hex(), scaled=hx(scaled), kept=[int(v) for v in kept]))
  x = noise(23, 5)
  for dfmt, order, size, pad in (("f", None, 8, 0.), ("f", ">", 8, 0.), ("f", "<", 5, -1.), ("d", None, 8, 0.),
                                 ("d", ">", 6, .25)):
    got = list(chunks.struct(list(x), size=size, dfmt=dfmt, byte_order=order, padval=pad))
    out["chunks"].append(dict(x=hx(x), dfmt=dfmt, byte_order=order, size=size, padval=hx(pad), ints=False,
                              chunks=[c.hex() for c in got]))
  xi = [rnd.randrange(-128, 128) for _ in range(19)]
  for dfmt, order, size, pad, scale in (("b", None, 4, 0, 1), ("h", "<", 8, -3, 250), ("h", ">", 8, 0, 250),
                                        ("i", None, 5, 7, 16000000), ("i", ">", 5, 0, 16000000),
                                        ("B", None, 4, 0, None), ("H", ">", 4, 9, None)):
    vals = [v * scale for v in xi] if scale else [abs(v) for v in xi]
    got = list(chunks.struct(list(vals), size=size, dfmt=dfmt, byte_order=order, padval=pad))
    out["chunks"].append(dict(x=vals, dfmt=dfmt, byte_order=order, size=size, padval=pad, ints=True,
                              chunks=[c.hex() for c in got]))
  # ParallelFilter: filters of different orders, memory shorter / longer than some of them,
  # a0 != 1, a pure gain and a FIR among the branches (lazy_filters.py:1048-1054)
  xs = noise(150, 31)
  branches = [lowpass.pole(800 * Hz), resonator.z_exp(1500 * Hz, 80 * Hz), ZFilter([.2, .3, .4], [2., -.5, .25]),
              ZFilter([.5]), 1 - .5 * z ** -3, resonator.poles_exp(300 * Hz, 20 * Hz)]
  for memory, zero in ((None, 0.), ([.7], 0.), ([.7, -.2], .1), ([.1, .2, .3, .4], -.05)):
    par = ParallelFilter(branches)
    kw = dict(zero=zero)
    if memory is not None:
      kw["memory"] = list(memory)
    out["parallel"].append(dict(sections=[dict(b=hx(f.numlist), a=hx(f.denlist)) for f in branches],
                                x=hx(xs), memory=None if memory is None else hx(memory), zero=hx(zero),
                                y=hx(list(par(list(xs), **kw)))))
  return out


# --------------------------------------------------------------------------
# 12. time-varying filters: Stream coefficients (lazy_filters.py:166-174, 197-224) and the
#     Stream-argument forms of the designs (:1179-1495)
# --------------------------------------------------------------------------
def timevar_cases():
  out = dict(direct=[], designs=[])
  rnd = random.Random(4242)
  N = 200
  x = [rnd.uniform(-1, 1) for _ in range(N)]
  ser = lambda lo, hi: [rnd.uniform(lo, hi) for _ in range(N)]
  b0, b2, a1, a2, a3, a0 = ser(.1, 1), ser(-1, 1), ser(-.9, .9), ser(-.4, .4), ser(-.2, .2), ser(.5, 2)

  def coefspec(lst):
    return [dict(series=hx(v)) if isinstance(v, list) else dict(const=hx(v)) for v in lst]

  def run(b, a, **kw):
    mk = lambda lst: [Stream(v) if isinstance(v, list) else v for v in lst]
    y = list(ZFilter(mk(b), mk(a))(list(x), **kw))
    out["direct"].append(dict(b=coefspec(b), a=coefspec(a), memory=hx(kw["memory"]) if "memory" in kw else None,
                              zero=hx(kw.get("zero", 0.)), y=hx(y)))
  run([b0], [1., a1])
  run([b0, 0.25, b2], [1., a1, .3, a3], memory=[.1, .2, .3], zero=.05)
  run([1., b2], [1., 0., a2], memory=[.7])
  run([b0, 0., 1.], [2., a1])
  run([b0, -1.], [-1., a1, a2], zero=-.1)
  run([b0, .5], [a0, a1, .2], memory=[.3], zero=.01)
  run([.5, b2, 0., 0., .1], [1., 0., 0., 0., a3])          # the wider register window (5 taps per side)
  run([b0], [1.])                                            # no recurrence at all
  # designs called with Streams
  cut = [(300 + 40 * k) * Hz for k in range(N)]
  frq = [(500 + 15 * k) * Hz for k in range(N)]
  bws = [(40 + .5 * k) * Hz for k in range(N)]
  specs = [("lowpass.pole", lowpass.pole, [cut]), ("lowpass.z", lowpass.z, [cut]),
           ("highpass.pole", highpass.pole, [cut]), ("highpass.z", highpass.z, [cut]),
           ("lowpass.pole_exp", lowpass.pole_exp, [cut]), ("highpass.z_exp", highpass.z_exp, [cut]),
           ("resonator.z_exp", resonator.z_exp, [frq, 80 * Hz]),
           ("resonator.poles_exp", resonator.poles_exp, [1000 * Hz, bws]),
           ("resonator.freq_z_exp", resonator.freq_z_exp, [frq, bws]),
           ("resonator.freq_poles_exp", resonator.freq_poles_exp, [frq, bws])]
  for name, fn, args in specs:
    mk = lambda: [Stream(v) if isinstance(v, list) else v for v in args]
    filt = fn(*mk())
    dump_coef = lambda lst: [dict(series=hx(list(v))) if isinstance(v, Stream) else dict(const=hx(v)) for v in lst]
    b_spec, a_spec = dump_coef(filt.numlist), dump_coef(filt.denlist)
    y = list(fn(*mk())(list(x)))
    out["designs"].append(dict(name=name, args=[hx(v) if isinstance(v, list) else dict(const=hx(v)) for v in args],
                               b=b_spec, a=a_spec, y=hx(y)))
  out["x"] = hx(x)
  return out


# --------------------------------------------------------------------------
# 13. time-variant filter *algebra* (the expressions of audiolazy/tests/test_filters.py:349-494):
#     what the reference's own Poly / ZFilter arithmetic on Stream coefficients produces
# --------------------------------------------------------------------------
def timevar_algebra_cases():
  from audiolazy import cycle, thub
  alpha = [-.5, -.2, -.1, 0, .1, .2, .5]
  data = [-7, 3] + list(range(10)) + [-50, 0] + list(range(70, -70, -11))
  delays = [1, 2, 3, 4]
  out = []
  for delay in (2, 3):
    gain1, gain2 = cycle(alpha), cycle(alpha[-2::-1])
    gain3, gain4, gain5, gain6 = Stream(1, 2, 3), Stream(.1, .7, -.5, -1e-3), Stream(.1, .2), Stream(3, 2, 1, 0)
    filt1 = (gain1.copy() * z ** -delay + gain2.copy() * z ** -delays[0]) / (1 + gain3.copy() * z ** -(delay + 2))
    filt2 = (gain4.copy() * z ** -delay + gain5.copy() * z ** -delays[-1]) / (1 + gain6.copy() * z ** -(delay - 1))
    out.append(dict(name="iir_sum", delay=delay, y=hx((filt1 + filt2)(cycle(data)).take(90))))
  for delay in (1, 4):
    gain1, gain2 = cycle([4, 5, 6, 5, 4, 3]), cycle(alpha[::-1])
    gain3, gain4 = Stream(*(alpha + [1, 2, 3])), Stream(.1, -.2, .3)
    gain5, gain6 = Stream(.1, .1, .1, -7), Stream(3, 2)
    filt1 = (gain1.copy() * z ** -delay + gain2.copy() * z ** -delays[0]) / (1 + gain3.copy() * z ** -(delay - 1))
    filt2 = (gain4.copy() * z ** -delay + gain5.copy() * z ** -delays[-1]) / (1 + gain6.copy() * z ** -(delay + 5))
    out.append(dict(name="iir_mul", delay=delay, y=hx((filt1 * filt2)(cycle(data)).take(90))))
  filt1 = (2 + Stream(1, 2, 3) * z ** -1) / Stream(1, 5)
  out.append(dict(name="copy", delay=0, y=hx(filt1.copy()(cycle(data[::-1])).take(50))))
  for delay in (1, 3):
    k = thub(min(alpha) + 2 + cycle(alpha), 3)
    filt = z ** -2 / k + Stream(5, 7) * z / (1 + z ** -delay)
    filt += filt.copy()
    filt *= z ** -1
    out.append(dict(name="sum_with_copy", delay=delay, y=hx(filt(cycle(data)).take(40))))
  a = Stream(1, 2, 3)
  out.append(dict(name="gain_in_denominator", delay=0, y=hx((1 / (a - z ** -1))(cycle(data)).take(50))))
  return out


# --------------------------------------------------------------------------
# 13. elementwise Stream stages around the filter: the operators of
#     lazy_stream.py:47-71, abs, clip (lazy_analysis.py:619-647), ** .5 / ** 2, and the
#     host-side callers maverage.deque (:523-566) and amdf (:677-716)
# --------------------------------------------------------------------------
def maps_cases():
  from audiolazy import clip, maverage, amdf, envelope
  inf, nan = float("inf"), float("nan")
  x = noise(300, 909) + [0.0, -0.0, 1.0, -1.0, 2.5, -2.5, 0.5, -0.5, inf, -inf, nan]
  y = noise(len(x), 910)                     # second operand: no zeros, no specials
  xfin = [v for v in x if v == v and abs(v) != inf]
  out = dict(x=hx(x), y=hx(y), xfin=hx(xfin), unary=[], scalar=[], binary=[], clip=[], callers=[])
  out["unary"].append(dict(op="abs", r=hx(list(abs(Stream(x))))))
  out["unary"].append(dict(op="neg", r=hx(list(-Stream(x)))))
  out["unary"].append(dict(op="sqrt", x=hx([abs(v) for v in xfin]), r=hx(list(Stream(abs(v) for v in xfin) ** .5))))
  out["unary"].append(dict(op="square_pow", x=hx(xfin), r=hx(list(Stream(xfin) ** 2))))
  for c in (0.37, -3.0, 1e-3):
    out["scalar"].append(dict(op="mul", c=hx(c), r=hx(list(Stream(x) * c))))
    out["scalar"].append(dict(op="rmul", c=hx(c), r=hx(list(c * Stream(x)))))
    out["scalar"].append(dict(op="add", c=hx(c), r=hx(list(Stream(x) + c))))
    out["scalar"].append(dict(op="sub", c=hx(c), r=hx(list(Stream(x) - c))))
    out["scalar"].append(dict(op="rsub", c=hx(c), r=hx(list(c - Stream(x)))))
    out["scalar"].append(dict(op="div", c=hx(c), r=hx(list(Stream(x) / c))))
    out["scalar"].append(dict(op="rdiv", c=hx(c), r=hx(list(c / Stream(y)))))   # (y has no zeros)
  out["binary"].append(dict(op="add", r=hx(list(Stream(x) + Stream(y)))))
  out["binary"].append(dict(op="sub", r=hx(list(Stream(x) - Stream(y)))))
  out["binary"].append(dict(op="mul", r=hx(list(Stream(x) * Stream(y)))))
  out["binary"].append(dict(op="div", r=hx(list(Stream(x) / Stream(y)))))
  for low, high in ((-1., 1.), (-.5, .5), (None, .3), (-.3, None), (None, None), (0., 0.), (-.25, 2.)):
    out["clip"].append(dict(low=None if low is None else hx(low), high=None if high is None else hx(high),
                            r=hx(list(clip(x, low, high)))))
  xs = noise(500, 911)
  out["xs"] = hx(xs)
  for size in (1, 4, 25):
    for zero in (0., .25):
      out["callers"].append(dict(fn="maverage.deque", size=size, zero=hx(zero),
                                 r=hx(list(maverage.deque(size)(xs, zero=zero)))))
  for lag, size in ((1, 4), (7, 16), (40, 10)):
    out["callers"].append(dict(fn="amdf", lag=lag, size=size, zero=hx(0.),
                               r=hx(list(amdf(lag, size)(xs)))))
  for strat in ("rms", "abs", "squared"):
    out["callers"].append(dict(fn="envelope." + strat, cutoff=hx(0.02),
                               r=hx(list(getattr(envelope, strat)(xs, 0.02)))))
  return out


# --------------------------------------------------------------------------
# 14. ``filt(other_zfilter)``: substitution of a ZFilter for z (lazy_filters.py:885-887,
#     ``sum(v * seq ** -k ...) / sum(v * seq ** -k ...)``) and what the constructor makes of a zero
#     denominator (lazy_filters.py:125-133: ``min()`` of no terms -> ValueError).  Seeded random rational
#     pairs: constants, pure delays / advances, integer coefficients, zero numerators.
# --------------------------------------------------------------------------
def typed(v):
  """A coefficient with its Python type: the reference keeps ints ints."""
  if isinstance(v, bool) or not isinstance(v, (int, float)):
    raise TypeError(type(v))
  return ["i", int(v)] if isinstance(v, int) else ["f", float(v).hex()]


def zfilter_outcome(fn):
  """What an expression gives: the (power, typed value) terms of numpoly / denpoly, or the exception's name."""
  try:
    r = fn()
  except Exception as exc:   # noqa: BLE001 -- the exception type IS the datum
    return dict(raises=type(exc).__name__)
  if not isinstance(r, ZFilter):
    return dict(value=typed(r))
  return dict(num=[[typed(k), typed(v)] for k, v in r.numpoly.terms()],
              den=[[typed(k), typed(v)] for k, v in r.denpoly.terms()])


def random_rational(rng, kind):
  """(num terms, den terms) as {power: value} dicts in z ** -1; ``kind`` picks the family."""
  def coef(integer):
    if integer:
      return rng.choice([-3, -2, -1, 1, 2, 3, 4])
    return rng.choice([rng.uniform(-2., 2.), round(rng.uniform(-3., 3.), 1), .5, -.25, 1.5])
  def poly(nterms, integer, lo, hi):
    return {p: coef(integer) for p in rng.sample(range(lo, hi + 1), nterms)}
  integer = rng.random() < .4
  if kind == "const":
    return {0: coef(integer)}, {0: 1}
  if kind == "delay":
    return {rng.choice([-3, -2, -1, 1, 2, 3]): rng.choice([1, 1, -1, coef(integer)])}, {0: 1}
  if kind == "zero":
    return {}, {0: 1} if rng.random() < .5 else poly(2, integer, 0, 2)
  if kind == "fir":
    return poly(rng.randint(1, 3), integer, -1, 3), {0: 1}
  num = poly(rng.randint(1, 3), integer, -1, 3)
  den = poly(rng.randint(1, 3), integer, 0, 3)
  if rng.random() < .3:
    den[0] = 1
  return num, den


def composition_cases():
  rng = random.Random(20260925)
  kinds = ["const", "delay", "zero", "fir", "rational", "rational", "rational", "fir"]
  out = []
  for idx in range(640):
    fk, gk = rng.choice(kinds), rng.choice(kinds)
    fnum, fden = random_rational(rng, fk)
    gnum, gden = random_rational(rng, gk)
    case = dict(f=[[[typed(k), typed(v)] for k, v in d.items()] for d in (fnum, fden)],
                g=[[[typed(k), typed(v)] for k, v in d.items()] for d in (gnum, gden)])
    def build(num, den):
      return ZFilter(dict(num), dict(den))
    case["f_built"] = zfilter_outcome(lambda: build(fnum, fden))
    case["g_built"] = zfilter_outcome(lambda: build(gnum, gden))
    case["f_of_g"] = zfilter_outcome(lambda: build(fnum, fden)(build(gnum, gden)))
    # division by / negative power of a filter whose numerator may be zero (empty denominator -> ValueError)
    case["g_over_f"] = zfilter_outcome(lambda: build(gnum, gden) / build(fnum, fden))
    case["k_over_f"] = zfilter_outcome(lambda: 2.5 / build(fnum, fden))
    case["f_inv"] = zfilter_outcome(lambda: build(fnum, fden) ** -1)
    out.append(case)
  return out


def surface_cases():
  """Attribute surface of Poly and of the filter objects (oracle/surface_probes.py), the reference's outcome of every
  probe on seeded random cases: {seed, n, rows} per family."""
  import surface_probes as sp
  return dict(poly=dict(seed=20260925, n=160, rows=sp.poly_outcomes(al, 20260925, 160)),
              filters=dict(seed=20260926, n=160, rows=sp.filter_outcomes(al, 20260926, 160)))


if __name__ == "__main__":
  print("audiolazy", al.__version__, "numpy", np.__version__)
  if len(sys.argv) > 1 and sys.argv[1] == "--only-surface":
    dump("surface.json", surface_cases())
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == "--only-maps":
    dump("maps.json", maps_cases())
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == "--only-generic":
    dump("generic_items.json", generic_item_cases())
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == "--only-composition":
    dump("composition.json", composition_cases())
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == "--only-lagrange":
    dump("lagrange.json", lagrange_cases())
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == "--only-covariance":
    dump("covariance.json", covariance_cases())
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == "--only-lpc-strategies":
    dump("lpc_strategies.json", lpc_strategy_cases())
    sys.exit(0)
  dump("filters.json", filt_cases())
  dump("lfilter_grid.json", lfilter_grid())
  dump("containers.json", container_cases())
  dump("designs.json", design_cases())
  dump("auditory.json", auditory_cases())
  dump("multichannel.json", multichannel_case())
  dump("lpc.json", lpc_cases())
  dump("blocks.json", blocks_cases())
  dump("karplus.json", karplus_case())
  dump("callers.json", callers_case())
  dump("formats.json", formats_cases())
  dump("timevar.json", timevar_cases())
  dump("timevar_algebra.json", timevar_algebra_cases())
  dump("maps.json", maps_cases())
  dump("lpc_strategies.json", lpc_strategy_cases())
  dump("generic_items.json", generic_item_cases())
  dump("composition.json", composition_cases())
  dump("surface.json", surface_cases())
  dump("covariance.json", covariance_cases())
  dump("lagrange.json", lagrange_cases())
