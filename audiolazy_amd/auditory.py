"""ERB bandwidths and gammatone filter designs (host-side, float64).

Mirror of the reference's auditory designs that feed the filter hot path
(reference audiolazy/lazy_auditory.py): ``erb`` (:55-88),
``gammatone_erb_constants`` (:91-125) and ``gammatone`` (:151-218).  A gammatone
band is a CascadeFilter of four two-pole sections; a *bank* of bands on many
input streams runs as one OUTER FilterBank on the GPU (:func:`gammatone_bank`).
"""
import math

from .filters import z, ZFilter, CascadeFilter, resonator, _accepts_streams
from .strategy import StrategyDict

__all__ = ["erb", "gammatone_erb_constants", "gammatone", "gammatone_bank", "erb_space"]

erb = StrategyDict("erb")


def _in_hz(freq, Hz):
  if Hz is None:
    if freq < 7:   # probably a value in rad/sample given without the Hz unit
      raise ValueError("Frequency out of range.")
    return freq, 1
  return freq / Hz, Hz


@erb.strategy("gm90", "glasberg_moore_90", "glasberg_moore")
def erb(freq, Hz=None):
  """Glasberg & Moore (1990) ERB, 24.7 (4.37e-3 f + 1) Hz (reference :55-70).
  ``erb(f_in_Hz)`` or ``erb(f * Hz, Hz)`` with the units of ``sHz``."""
  fHz, unit = _in_hz(freq, Hz)
  return 24.7 * (4.37e-3 * fHz + 1.) * unit


@erb.strategy("mg83", "moore_glasberg_83")
def erb(freq, Hz=None):
  """Moore & Glasberg (1983) ERB, 6.23e-6 f^2 + 93.39e-3 f + 28.52 Hz (reference :73-88)."""
  fHz, unit = _in_hz(freq, Hz)
  return (6.23e-6 * fHz ** 2 + 93.39e-3 * fHz + 28.52) * unit


erb.default = erb.gm90


def gammatone_erb_constants(n):
  """(1/a_n, c_n) of Holdsworth et al. for an order-n gammatone (reference :91-125):
  the first scales an ERB into the gammatone bandwidth parameter, the product of
  both into the 3 dB bandwidth."""
  tnt = 2 * n - 2
  return (math.factorial(n - 1) ** 2 / (math.pi * math.factorial(tnt) * 2 ** -tnt),
          2 * (2 ** (1. / n) - 1) ** .5)


gammatone = StrategyDict("gammatone")


@gammatone.strategy("sampled")
@_accepts_streams
def gammatone(freq, bandwidth, phase=0, eta=4):
  """Impulse-invariant ("sampled") gammatone, n^(eta-1) e^(-bandwidth n) cos(freq n + phase)
  (reference :158-182): the (eta-1)-th z-derivative of the one-pole-pair kernel, split
  into one section carrying the whole numerator and eta-1 all-pole sections, each with
  unit gain at ``freq``."""
  A = math.exp(-bandwidth)
  numerator = math.cos(phase) - A * math.cos(freq - phase) * z ** -1
  denominator = 1 - 2 * A * math.cos(freq) * z ** -1 + A ** 2 * z ** -2
  filt = (numerator / denominator).diff(n=eta - 1, mul_after=-z)
  f0 = ZFilter(filt.numpoly) / denominator
  f0 = f0 / abs(f0.freq_response(freq))
  fn = 1 / denominator
  fn = fn / abs(fn.freq_response(freq))
  return CascadeFilter([f0] + [fn] * (eta - 1))


@gammatone.strategy("slaney")
@_accepts_streams
def gammatone(freq, bandwidth):
  """Slaney's cascade of four pole pairs, each with one real zero (reference :188-202)."""
  A = math.exp(-bandwidth)
  cosw, sinw = math.cos(freq), math.sin(freq)
  sig = [1., -1.]
  coeff = [cosw + s1 * (math.sqrt(2) + s2) * sinw for s1 in sig for s2 in sig]
  denominator = 1 - 2 * A * cosw * z ** -1 + A ** 2 * z ** -2
  sections = [(1 - A * c * z ** -1) / denominator for c in coeff]
  return CascadeFilter(f / abs(f.freq_response(freq)) for f in sections)


@gammatone.strategy("klapuri")
def gammatone(freq, bandwidth):
  """Klapuri's cascade: resonator.z_exp, resonator.poles_exp, twice, at twice the
  bandwidth (reference :208-218).  Numbers or Streams (the resonators take both)."""
  from .stream import thub
  bw2 = thub(thub(bandwidth, 1) * 2, 4)
  freq = thub(freq, 4)
  return CascadeFilter(reson(freq, bw2) for reson in [resonator.z_exp, resonator.poles_exp] * 2)


gammatone.default = gammatone.sampled


def erb_space(low, high, n, Hz=1.):
  """n centre frequencies (same unit as ``low``/``high``) uniform on the ERB-rate
  scale of Glasberg & Moore, E(f) = 21.4 log10(4.37e-3 f + 1).  Not in the reference
  (it has no bank helper); BASELINE config 4 spaces its bands this way."""
  def rate(f):
    return 21.4 * math.log10(4.37e-3 * (f / Hz) + 1.)
  lo, hi = rate(low), rate(high)
  return [((10 ** ((lo + (hi - lo) * i / max(n - 1, 1)) / 21.4) - 1.) / 4.37e-3) * Hz for i in range(n)]


def gammatone_bank(freqs, n_inputs, strategy="slaney", Hz=1., device=0):
  """A gammatone filterbank as one GPU bank: every band on every input stream.

  freqs : centre frequencies in rad/sample when ``Hz`` is the ``sHz`` unit, with the
          canonical bandwidth of the reference's example,
          ``gammatone_erb_constants(4)[0] * erb(fc, Hz)`` (examples/gammatone_plots.py:47).
  Returns a FilterBank in OUTER mode: output channel = band * n_inputs + input.
  """
  from .bank import FilterBank
  design = getattr(gammatone, strategy)
  k = gammatone_erb_constants(4)[0]
  bands = [design(fc, k * erb(fc, Hz)) for fc in freqs]
  return FilterBank.from_filters(bands, n_inputs=n_inputs, mode="outer", device=device)
