"""Signal sources the filter path is fed with in the reference's own examples.

Host-side mirror of ``white_noise`` (reference audiolazy/lazy_synth.py:394-415, the cfg1 /
benchmark input source), ``zeros`` (:303-322) and ``karplus_strong`` (:624-657, a feedback comb
at a fractional period run on silence with noise as its memory).  The sources are plain Python
generators wrapped in a Stream; karplus_strong's filtering runs on the GPU engine.
"""
import math
import random

from .stream import Stream, rint
from .filters import comb


def white_noise(dur=None, low=-1., high=1.):
  """Uniform noise in [low, high]: ``dur`` samples, endless when None / inf."""
  def gen():
    if dur is None or (isinstance(dur, float) and math.isinf(dur) and dur > 0):
      while True:
        yield random.uniform(low, high)
    for _ in range(rint(dur)):
      yield random.uniform(low, high)
  return Stream(gen())


def _constant(value, dur):
  def gen():
    if dur is None or (isinstance(dur, float) and math.isinf(dur) and dur > 0):
      while True:
        yield value
    for _ in range(rint(dur)):
      yield value
  return Stream(gen())


def zeros(dur=None):
  """``dur`` float zeros, endless when None / inf (reference :303-322)."""
  return _constant(0., dur)


zeroes = zeros


def ones(dur=None):
  """``dur`` float ones, endless when None / inf (reference :280-300)."""
  return _constant(1., dur)


def karplus_strong(freq, tau=2e4, memory=white_noise):
  """Karplus-Strong plucked string: ``comb.tau(2 pi / freq, tau).linearize()`` applied to
  silence, the delay line pre-loaded by ``memory`` (a callable given the memory size, or an
  iterable).  ``freq`` in rad/sample, ``tau`` in samples.  Returns an endless Stream."""
  return comb.tau(2 * math.pi / freq, tau).linearize()(zeros(), memory=memory)
