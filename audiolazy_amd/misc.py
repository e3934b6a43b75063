"""Small helpers the reference's filter examples and tests lean on (host side, a few lines each):
``dB10`` / ``dB20`` (reference audiolazy/lazy_math.py:112-125), ``freq2lag`` / ``lag2freq``
(lazy_misc.py:323-331), ``almost_eq`` (lazy_misc.py:232-296) and ``line`` (lazy_synth.py:143-180).
"""
import math

from .stream import Stream, rint
from .strategy import StrategyDict


def _elementwise(fn):
  def mapped(data):
    if hasattr(data, "__iter__"):
      out = (fn(v) for v in data)
      return Stream(out) if isinstance(data, Stream) else type(data)(out)
    return fn(data)
  mapped.__name__ = fn.__name__
  mapped.__doc__ = fn.__doc__
  return mapped


@_elementwise
def dB10(data):
  """Power ratio (a squared amplitude) in dB; -inf for zero."""
  return 10 * math.log10(abs(data)) if data != 0 else -float("inf")


@_elementwise
def dB20(data):
  """Amplitude ratio in dB; -inf for zero."""
  return 20 * math.log10(abs(data)) if data != 0 else -float("inf")


def freq2lag(v):
  """Frequency in rad/sample -> period in samples (and back: the map is its own inverse)."""
  return 2 * math.pi / v


lag2freq = freq2lag
freq_to_lag = lag_to_freq = freq2lag          # (the reference's deprecated names, lazy_misc.py:326, 332)

DEFAULT_SAMPLE_RATE = 44100                   # Hz (lazy_misc.py:41)


def cached(func):
  """Memoise a function of positional arguments; the results live in ``f.cache``, a dict keyed by the argument tuple
  (lazy_misc.py:335-349)."""
  import functools

  class Cache(dict):
    def __missing__(self, key):
      self[key] = result = func(*key)
      return result
  cache = Cache()
  f = functools.wraps(func)(lambda *key: cache[key])
  f.cache = cache
  return f


def elementwise(name="", pos=None):
  """Decorator factory: the decorated function maps over ONE of its arguments (keyword ``name`` and / or position
  ``pos``; the first positional one by default) when that argument is an iterable -- a generator gives a generator, a
  NumPy array an array, a Stream a Stream, any other container its own type (lazy_misc.py:163-228)."""
  import functools
  import itertools
  import types
  if name == "" and pos is None:
    pos = 0

  def decorator(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
      positional = pos is not None and pos < len(args)
      arg = args[pos] if positional else kwargs[name]
      if hasattr(arg, "__iter__") and not isinstance(arg, (str, bytes)):
        if positional:
          data = (func(*(args[:pos] + (v,) + args[pos + 1:]), **kwargs) for v in arg)
        else:
          data = (func(*args, **dict(kwargs, **{name: v})) for v in arg)
        if isinstance(arg, (types.GeneratorType, range, enumerate, zip, itertools.zip_longest, map, filter)):   # (lazy_compat.py:52-53)
          return data
        kind = type(arg)
        if getattr(kind, "__module__", None) == "numpy":
          import numpy as np
          return (np.array if kind.__name__ == "ndarray" else np.asmatrix)(list(data))
        if issubclass(kind, Stream):
          return Stream(data)
        return kind(data)
      return func(*args, **kwargs)
    return wrapper
  return decorator


def _pairs(a, b, pad):
  import itertools
  return itertools.zip_longest(a, b, fillvalue=pad)


almost_eq = StrategyDict("almost_eq")


@almost_eq.strategy("bits")
def almost_eq(a, b, bits=32, tol=1, ignore_type=True, pad=0.):
  """``a == b`` up to the last ``tol`` bits of a ``bits``-wide float's significand, item by item
  through nested iterables (the reference's default comparison helper)."""
  if not (ignore_type or type(a) == type(b)):
    return False
  it_a, it_b = hasattr(a, "__iter__"), hasattr(b, "__iter__")
  if it_a != it_b:
    return False
  if it_a:
    return all(almost_eq.bits(x, y, bits, tol, ignore_type) for x, y in _pairs(a, b, pad))
  significand = {32: 23, 64: 52, 80: 63, 128: 112}[bits]
  return abs(a - b) <= 2 ** (tol - significand - 1) * abs(a + b)


@almost_eq.strategy("diff")
def almost_eq(a, b, max_diff=1e-7, ignore_type=True, pad=0.):
  """``|a - b| <= max_diff`` item by item through nested iterables."""
  if not (ignore_type or type(a) == type(b)):
    return False
  it_a, it_b = hasattr(a, "__iter__"), hasattr(b, "__iter__")
  if it_a != it_b:
    return False
  if it_a:
    return all(almost_eq.diff(x, y, max_diff, ignore_type) for x, y in _pairs(a, b, pad))
  return abs(a - b) <= max_diff


def line(dur, begin=0., end=1., finish=False):
  """Finite Stream of ``dur`` samples on the straight line from ``begin`` towards ``end``;
  ``end`` itself is the last sample only with ``finish=True``."""
  def gen():
    m = (end - begin) / (dur - (1. if finish else 0.))
    for sample in range(rint(dur)):
      yield begin + sample * m
  return Stream(gen())
