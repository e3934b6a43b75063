"""Callers of the filter path in the reference's analysis module.

Host-side mirror of ``envelope`` (reference audiolazy/lazy_analysis.py:440-520), of the
``maverage`` strategies (:523-616; ``deque`` is the reference's default and is a plain generator,
``recursive`` / ``fir`` are filters) and of ``amdf`` (:677-716, a comb difference, ``abs`` and a
moving average).  The elementwise pre/post stages are plain
lazy Stream operations (the same CPython float operations as the reference, so the results
are identical); the lowpass / moving-average filter in the middle runs on the GPU engine.
"""
import collections
import math

from .filters import z, lowpass
from .strategy import StrategyDict
from .stream import Stream

__all__ = ["envelope", "maverage", "amdf"]

envelope = StrategyDict("envelope")


@envelope.strategy("rms")
def envelope(sig, cutoff=math.pi / 512):
  """Root of the lowpassed squared signal (reference :440-465)."""
  return lowpass(cutoff)(Stream(sig) ** 2) ** .5


@envelope.strategy("abs")
def envelope(sig, cutoff=math.pi / 512):
  """Lowpassed absolute value (reference :468-493)."""
  return lowpass(cutoff)(abs(Stream(sig)))


@envelope.strategy("squared")
def envelope(sig, cutoff=math.pi / 512):
  """Lowpassed squared signal (reference :496-520)."""
  return lowpass(cutoff)(Stream(sig) ** 2)


envelope.default = envelope.rms

maverage = StrategyDict("maverage")


@maverage.strategy("deque")
def maverage(size):
  """Moving average kept as a running sum over a deque of ``x / size`` terms (reference :525-559):
  a callable ``(sig, zero=0.) -> Stream``.  Not a filter object (no algebra, no frequency
  response) and, being a running sum, not the same roundings as the filter strategies; it runs
  on the host exactly like the reference's."""
  size_inv = 1. / size

  def maverage_filter(sig, zero=0.):
    def gen():
      data = collections.deque((zero * size_inv for _ in range(size)), maxlen=size)
      mean_value = zero
      for el in sig:
        mean_value -= data.popleft()
        new_value = el * size_inv
        data.append(new_value)
        mean_value += new_value
        yield mean_value
    return Stream(gen())
  return maverage_filter


@maverage.strategy("recursive", "feedback")
def maverage(size):
  """Moving average as the recursive filter (1/size)(1 - z**-size)/(1 - z**-1) (reference :569-591)."""
  return (1. / size) * (1 - z ** -size) / (1 - z ** -1)


@maverage.strategy("fir")
def maverage(size):
  """Moving average as a ``size``-tap FIR of 1/size (reference :594-616)."""
  return sum((1. / size) * z ** -i for i in range(size))


maverage.default = maverage.deque


def amdf(lag, size):
  """Average Magnitude Difference Function for a fixed lag (reference :677-716): the comb
  difference ``(1 - z ** -lag).linearize()`` on the GPU engine, ``abs``, then ``maverage(size)``.
  Returns a callable ``(sig, zero=0.) -> Stream``."""
  filt = (1 - z ** -lag).linearize()

  def amdf_filter(sig, zero=0.):
    return maverage(size)(abs(filt(sig, zero=zero)), zero=zero)
  return amdf_filter
