"""The reference's TestResonator / TestLowpassHighpass (audiolazy/tests/test_filters.py:569-663)
and test_auditory.py (:37-92) restated with audiolazy_amd.  Design-time only: no GPU."""
import itertools
import math

import pytest

from audiolazy_amd import (resonator, lowpass, highpass, ZFilter, CascadeFilter, Stream, repeat, erb,
                           gammatone_erb_constants, gammatone, sHz)
from test_reference_algebra import almost_eq

p = pytest.mark.parametrize
pi = math.pi


def close(a, b, tol=1):
  """almost_eq.bits on numbers (lazy_misc.py:232-266): |a - b| <= 2 ** (tol - 24) * |a + b|."""
  return abs(a - b) <= 2 ** (tol - 24) * abs(a + b)


def dB10(v):
  return 10 * math.log10(v)


def dB20(v):
  return 20 * math.log10(abs(v))


def line(dur, begin, end):
  return [begin + (end - begin) * i / dur for i in range(dur)]


# ------------------------------------------------------------------ TestResonator (:569-599)
@p("func", list(resonator))
def test_zeros_and_number_of_poles(func):
  names = set(func.__name__.split("_"))
  filt = func(pi / 2, pi / 18)
  assert isinstance(filt, ZFilter) and len(filt.denominator) == 3
  num = filt.numerator
  if "z" in names:
    assert len(num) == 3 and num[1] == 0 and num[0] == -num[2]
  if "poles" in names:
    assert len(num) == 1


@p("func", list(resonator))
@p("freq", [pi / 2, pi / 3, 2 * pi / 3])
@p("bw", [pi * k / 15 for k in range(1, 5)])
def test_radius_range(func, freq, bw):
  assert 0 < func(freq, bw).denominator[2] < 1


@p("func", [r for r in resonator if "freq" not in r.__name__.split("_")])
@p("freq", [pi * k / 7 for k in range(1, 7)])
@p("bw", [pi / 25, pi / 30])
def test_gain_0dB_at_given_freq(func, freq, bw):
  assert abs(dB20(func(freq, bw).freq_response(freq))) <= 5e-14


# ------------------------------------------------------------ TestLowpassHighpass (:602-663)
@p("filt_func", [lowpass.pole, highpass.pole, lowpass.z, highpass.z])
@p("freq", [pi * k / 7 for k in range(1, 7)])
def test_3dB_gain(filt_func, freq):
  assert close(dB20(filt_func(freq).freq_response(freq)), dB10(.5))


@p("sdict", [lowpass, highpass])
@p(("freq", "tol"), list(zip([pi / 300, pi / 30, pi / 15, pi / 10, 2 * pi / 15, pi / 6], [7, 13, 15, 16, 17, 18])))
def test_pole_exp_for_small_cutoff_frequencies(sdict, freq, tol):
  if sdict is highpass:
    freq = pi - freq
  filt, expected = sdict.pole_exp(freq), sdict.pole(freq)
  assert all(close(a, b, tol) for a, b in zip(filt.numerator, expected.numerator))
  assert all(close(a, b, tol) for a, b in zip(filt.denominator, expected.denominator))
  assert close(abs(filt.freq_response(freq)), .5 ** .5, tol)
  assert abs(dB20(filt.freq_response(freq)) - dB10(.5)) <= .1


@p("filt_func", list(lowpass))
@p("freq", [pi * k / 7 for k in range(1, 7)])
def test_lowpass_is_lowpass(filt_func, freq):
  filt = filt_func(freq)
  assert close(abs(filt.freq_response(0.)), 1.)
  freqs = Stream(line(50, 0, pi))
  for a, b in filt.freq_response(freqs).map(abs).blocks(size=2, hop=1):
    assert b < a


@p("filt_func", list(highpass))
@p("freq", [pi * k / 7 for k in range(1, 7)])
def test_highpass_is_highpass(filt_func, freq):
  filt = filt_func(freq)
  assert close(abs(filt.freq_response(pi)), 1.)
  freqs = Stream(line(50, 0, pi))
  for a, b in filt.freq_response(freqs).map(abs).blocks(size=2, hop=1):
    assert a < b


@p(("filt_func", "fsign"), [(lowpass.z, 1), (highpass.z, -1)])
def test_single_zero_strategies_zeroed_R_denominator_lti(filt_func, fsign):
  filt = filt_func(pi / 2)
  assert filt.denpoly[0] == 1 and abs(filt.denpoly[1]) <= 3e-16
  assert almost_eq(filt.numlist, [.5, fsign * .5])


@p(("filt_func", "fsign"), [(lowpass.z, 1), (highpass.z, -1)])
def test_single_zero_strategies_zeroed_R_denominator_tvar(filt_func, fsign):
  filt = filt_func(repeat(pi / 2))
  assert filt.denpoly[0] == 1
  pole_sig = filt.denpoly[1]
  num_sig0, num_sig1 = filt.numlist
  n = 3
  assert all(abs(v) <= 3e-16 for v in pole_sig.limit(n))
  assert almost_eq(num_sig0.limit(n), [.5] * n)
  assert almost_eq(num_sig1.limit(n), [fsign * .5] * n)


# ------------------------------------------------------------------ test_auditory.py (:37-92)
@p(("freq", "bandwidth"), [(1000, 132.639), (3000, 348.517)])
def test_glasberg_moore_slaney_example(freq, bandwidth):
  assert abs(erb["gm90"](freq) - bandwidth) <= 5e-4


@p("erb_func", list(erb))
@p("rate", [8000, 22050, 44100])
@p("freq", [440, 20, 2e4])
def test_erb_two_input_methods(erb_func, rate, freq):
  Hz = sHz(rate)[1]
  assert close(erb_func(freq) * Hz, erb_func(freq * Hz, Hz))
  if freq < rate:
    with pytest.raises(ValueError):
      erb_func(freq * Hz)


@p(("n", "an", "aninv", "cn", "cninv"),
   [(1, 3.142, 0.318, 2.000, 0.500), (2, 1.571, 0.637, 1.287, 0.777), (3, 1.178, 0.849, 1.020, 0.981),
    (4, 0.982, 1.019, 0.870, 1.149), (5, 0.859, 1.164, 0.771, 1.297), (6, 0.773, 1.293, 0.700, 1.429),
    (7, 0.709, 1.411, 0.645, 1.550), (8, 0.658, 1.520, 0.602, 1.662), (9, 0.617, 1.621, 0.566, 1.767)])
def test_annex_c_table_1(n, an, aninv, cn, cninv):
  x, y = gammatone_erb_constants(n)
  assert abs(x - aninv) <= 5e-4 and abs(y - cn) <= 5e-4
  assert abs(1. / x - an) <= 5e-4 and abs(1. / y - cninv) <= 5e-4


some_data = [lambda: pi / 7, lambda: Stream(0, 1, 2, 1), lambda: [pi / 3, pi / 4, pi / 5, pi / 6]]


@p(("filt_func", "freq", "bw"),
   [(gf, lambda: pi / 5, lambda: pi / 19) for gf in gammatone] +
   [(gammatone.klapuri, freq, bw) for freq, bw in itertools.product(some_data, some_data)])
def test_number_of_poles_order(filt_func, freq, bw):
  cfilt = filt_func(freq=freq(), bandwidth=bw())
  assert isinstance(cfilt, CascadeFilter) and len(cfilt) == 4
  for filt in cfilt:
    assert len(filt.denominator) == 3


# ------------------------------------------- test_filters_extdep.py:239-266 (TestResonatorScipy)
@p("func", list(resonator))
@p("freq", [pi * k / 9 for k in range(1, 9)])
@p("bw", [pi / 23, pi / 31])
def test_max_gain_is_at_resonance(func, freq, bw):
  from scipy.optimize import fminbound
  names = func.__name__.split("_")
  filt = func(freq, bw)
  resonance_freq = fminbound(lambda x: -dB20(filt.freq_response(x)), 0, pi, xtol=1e-10)
  assert abs(dB20(filt.freq_response(resonance_freq))) <= 1e-12
  if "freq" in names:                           # the given frequency is the pole angle
    R = math.sqrt(filt.denominator[2])
    assert 0 < R < 1
    cosf = math.cos(freq)
    assert close(cosf, -filt.denominator[1] / (2 * R))
    cosw = cosf * (2 * R) / (1 + R ** 2) if "z" in names else cosf * (1 + R ** 2) / (2 * R)
    assert close(cosw, math.cos(resonance_freq))
  else:                                         # the given frequency is the resonance frequency
    assert close(freq, resonance_freq)
