"""Callers of the filter path in the reference's analysis module.

Host-side mirror of ``envelope`` (reference audiolazy/lazy_analysis.py:440-520) and of the
filter-shaped ``maverage`` strategies (:569-616).  The elementwise pre/post stages are plain
lazy Stream operations (the same CPython float operations as the reference, so the results
are identical); the lowpass / moving-average filter in the middle runs on the GPU engine.
"""
import math

from .filters import z, lowpass
from .strategy import StrategyDict
from .stream import Stream

__all__ = ["envelope", "maverage"]

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


@maverage.strategy("recursive", "feedback")
def maverage(size):
  """Moving average as the recursive filter (1/size)(1 - z**-size)/(1 - z**-1) (reference :569-591)."""
  return (1. / size) * (1 - z ** -size) / (1 - z ** -1)


@maverage.strategy("fir")
def maverage(size):
  """Moving average as a ``size``-tap FIR of 1/size (reference :594-616)."""
  return sum((1. / size) * z ** -i for i in range(size))


maverage.default = maverage.recursive   # (the reference's default, a deque generator, is not a filter)
