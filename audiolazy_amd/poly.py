"""Sparse Laurent polynomial in one variable: the algebra under ``z ** -1``.

Host-side mirror of what the filter hot path needs from the reference's ``Poly``
(reference audiolazy/lazy_poly.py:66-490): construction from list / dict /
scalar, ``+ - * / **``, composition, ``diff``, ``terms`` / ``values`` /
``order``.  Coefficients are plain Python numbers or Streams (a Stream coefficient
makes the filter time-varying; whenever an operation reads one to build a new
coefficient it reads a ``copy()``, so the operand stays usable -- the job of the
reference's ``thub`` calls, lazy_poly.py:392-395).

Rounding contract: filter *design* results must equal the reference's to the
last bit, and they depend on evaluation order, so the two places where order
matters follow the reference exactly --
  * ``a * b`` accumulates ``new[k1 + k2] += v1 * v2`` with ``a``'s terms in the
    outer loop and ``b``'s in the inner one, both in insertion order
    (lazy_poly.py:388-402);
  * ``a + b`` keeps ``a``'s term order, then ``b``'s new powers, and sums shared
    powers as ``a_k + b_k`` (lazy_poly.py:373-380).
Terms whose coefficient equals zero are dropped on construction (:136-143).
"""
import numbers

from .stream import Stream, IGNORED_CLASSES


def _use(v):
  """The value of a coefficient for building another one (a Stream is teed, not consumed)."""
  return v.copy() if isinstance(v, Stream) else v


class Poly(object):
  __slots__ = ("_t", "_zero", "_hash")

  def __init__(self, data=None, zero=None):
    self._zero = 0. if zero is None else zero
    if data is None:
      terms = {}
    elif isinstance(data, Poly):
      terms = dict(data._t)
      if zero is None:
        self._zero = data._zero
    elif isinstance(data, dict):
      terms = dict(data)
    elif isinstance(data, list):
      terms = dict(enumerate(data))
    else:
      terms = {0: data}
    # compaction, in the reference's order (lazy_poly.py:131-139): a float power that is an integer becomes that
    # integer and moves to the END of the creation order; a coefficient equal to ``zero`` is dropped (a Stream never is)
    for power, coef in list(terms.items()):
      if isinstance(power, float) and power.is_integer():
        del terms[power]
        power = int(round(power))
        terms[power] = coef
      if (not isinstance(coef, Stream)) and coef == self._zero:
        del terms[power]
    self._t = terms

  # -- the value a missing power has (lazy_poly.py:141-151) ----------------------
  @property
  def zero(self):
    return self._zero

  @zero.setter
  def zero(self, value):
    if hasattr(self, "_hash"):
      raise TypeError("Used this Poly instance as a hashable before")
    self._zero = value
    for power, coef in list(self._t.items()):
      if (not isinstance(coef, Stream)) and coef == value:
        del self._t[power]

  # -- views -----------------------------------------------------------------
  def terms(self, sort="auto", reverse=False):
    """(power, coefficient) pairs: sorted by power for integer powers (``sort="auto"``), in creation order
    otherwise; ``reverse`` reverses either order (lazy_poly.py:170-199)."""
    if sort == "auto":
      sort = self.is_laurent()
    if sort:
      keys = sorted(self._t, reverse=reverse)
    elif reverse:
      keys = list(reversed(list(self._t)))
    else:
      keys = list(self._t)
    return ((k, self._t[k]) for k in keys)

  def values(self):
    """Dense coefficient list for powers 0 .. order (lazy_poly.py:159-168)."""
    if not self._t:
      return []
    return [self[k] for k in range(self.order + 1)]

  @property
  def order(self):
    if not self.is_polynomial():
      raise AttributeError("Power needs to be positive integers")
    return max(self._t) if self._t else 0

  @property
  def roots(self):
    """All roots, by numpy.roots on the dense coefficient list like the reference (lazy_poly.py:481-487)."""
    import numpy as np
    return np.roots(list(self.values())[::-1]).tolist()

  def is_polynomial(self):
    return all(isinstance(k, numbers.Integral) and not isinstance(k, bool) and k >= 0 for k in self._t)

  def is_laurent(self):
    """Integer powers only, negative ones allowed (lazy_poly.py:212-231)."""
    return all(isinstance(k, numbers.Integral) and not isinstance(k, bool) for k in self._t)

  def __len__(self):
    return len(self._t)

  def __getitem__(self, power):
    return self._t[power] if power in self._t else self._zero

  def __setitem__(self, power, coef):
    """Allowed until the instance has been hashed (lazy_poly.py:357-367)."""
    if getattr(self, "_hash", False):
      raise TypeError("Used this Poly instance as a hashable before")
    if isinstance(power, float) and power.is_integer():
      power = int(round(power))
    if isinstance(coef, Stream) or coef != self._zero:
      self._t[power] = coef
    elif power in self._t:
      del self._t[power]

  def __eq__(self, other):
    if not isinstance(other, Poly):
      other = Poly(other, zero=self._zero)

    def same(v, w):
      if isinstance(v, Stream) or isinstance(w, Stream):   # Streams compare by identity
        return v is w
      return v == w
    return same(self._zero, other._zero) and len(self._t) == len(other._t) and \
        all(k in other._t and same(v, other._t[k]) for k, v in self._t.items())

  def __ne__(self, other):
    return not self == other

  def __hash__(self):
    if not hasattr(self, "_hash"):                          # (from here on the instance is immutable, :153-156)
      self._hash = hash((frozenset(self._t.items()), self._zero))
    return self._hash

  def copy(self, zero=None):
    """Same terms; Stream coefficients are teed so both polynomials stay usable (:255-263)."""
    return Poly({k: _use(v) for k, v in self._t.items()}, zero=self._zero if zero is None else zero)

  # -- ring operations ---------------------------------------------------------
  # (results carry the LEFT operand's ``zero``; a number on the left is cast with the Poly's, lazy_poly.py:50-62)
  def __neg__(self):
    return Poly({k: -_use(v) for k, v in self._t.items()}, zero=self._zero)

  def __pos__(self):
    return Poly({k: +_use(v) for k, v in self._t.items()}, zero=self._zero)

  def __add__(self, other):
    if not isinstance(other, Poly):
      other = Poly(other)
    out = {k: _use(v) for k, v in self._t.items()}
    for k, v in other._t.items():
      out[k] = (out[k] + _use(v)) if k in out else _use(v)
    return Poly(out, zero=self._zero)

  def __radd__(self, other):
    return Poly(other, zero=self._zero) + self

  def __sub__(self, other):
    return self + (-(other if isinstance(other, Poly) else Poly(other)))

  def __rsub__(self, other):
    return Poly(other, zero=self._zero) - self

  def __mul__(self, other):
    if not isinstance(other, Poly):
      other = Poly(other)
    out = {}
    for k1, v1 in self._t.items():
      for k2, v2 in other._t.items():
        k = k1 + k2
        if k in out:
          out[k] += _use(v1) * _use(v2)
        else:
          out[k] = _use(v1) * _use(v2)
    return Poly(out, zero=self._zero)

  def __rmul__(self, other):
    return Poly(other, zero=self._zero) * self

  def __truediv__(self, other):
    if isinstance(other, Poly):
      if len(other) == 0:
        raise ZeroDivisionError("Dividing Poly instance by zero")
      if len(other) != 1:
        raise NotImplementedError("Can't divide general Poly instances")
      (shift, value), = other._t.items()
      return Poly({k - shift: _use(v) / _use(value) for k, v in self._t.items()}, zero=self._zero)
    return Poly({k: _use(v) / _use(other) for k, v in self._t.items()}, zero=self._zero)

  def __pow__(self, n):
    if isinstance(n, Poly):
      if any(k != 0 for k in n._t):
        raise NotImplementedError("Can't power general Poly instances")
      n = n[0]
    if n == 0:
      return Poly(1, zero=self._zero)
    if len(self._t) == 0:
      return Poly(zero=self._zero)
    if len(self._t) == 1:
      (k, v), = self._t.items()
      if isinstance(v, Stream):
        return Poly({k * n: _use(v) ** n}, zero=self._zero)
      return Poly({k * n: 1 if v == 1 else v ** n}, zero=self._zero)   # lazy_poly.py:445-449
    # ((p * p) * p) ... over n - 1 copies and the instance itself, exactly as the reference's reduce over a list
    # builds it (lazy_poly.py:450): a non-integer n is a TypeError there, n <= 1 gives the instance itself
    factors = [self.copy()] * (n - 1) + [self]
    out = factors[0]
    for f in factors[1:]:
      out = out * f
    return out

  # -- calculus / evaluation -----------------------------------------------------
  def diff(self, n=1):
    """n-th derivative (lazy_poly.py:259-266)."""
    terms = self._t
    for _ in range(n):
      terms = {k - 1: k * v for k, v in terms.items() if k != 0}
    return Poly(terms, zero=self._zero)

  def integrate(self):
    """Antiderivative without a constant (lazy_poly.py:274-282)."""
    if -1 in self._t:
      raise ValueError("Unable to integrate term that powers to -1")
    return Poly({k + 1: v / (k + 1) for k, v in self._t.items()}, zero=self._zero)

  def __call__(self, value, horner="auto"):
    """Evaluate at a number or a Stream (elementwise), or substitute another Poly / algebraic
    object.

    Substitution is ``sum(coeff * value ** power)`` over the terms in creation order
    (lazy_poly.py:313-316); numbers and Streams use the Horner-like scheme -- merged steps for missing powers --
    when ``horner`` says so (``"auto"``: for plain polynomials) and the direct sum otherwise (:318-349).  A Stream is
    teed once per use (``thub``), so it may be any single-pass iterable.
    """
    from .stream import thub
    # (hasattr is useless on a Stream: attribute access is elementwise there)
    if isinstance(value, Poly) or (not isinstance(value, Stream) and hasattr(value, "numpoly")):
      total = 0
      for power, coef in self._t.items():
        total = total + _use(coef) * value ** power
      return Poly(total, self._zero) if isinstance(value, Poly) else total
    if not self._t:
      return self._zero
    if not isinstance(value, Stream) and not hasattr(value, "__iter__"):
      if value is None:
        raise TypeError("cannot evaluate a non-empty Poly at None")
      if value == 0:
        return self[0]
    value = thub(value, len(self._t))
    if horner == "auto":
      horner = self.is_polynomial()
    if horner:
      try:
        pairs = [(k, _use(v)) for k, v in self.terms(sort=True, reverse=True)]
      except TypeError:                                   # powers without an order
        raise ValueError("Can't apply Horner-like scheme")
      last_power, result = pairs[0]
      for power, coef in pairs[1:]:
        gap = last_power - power
        result = coef + result * (value if gap == 1 else value ** gap)
        last_power = power
      return result * value ** last_power
    total = 0
    for power, coef in self.terms():
      total = total + _use(coef) * value ** power
    return total

  def __str__(self):
    """``7 - x + x^5``: the reference's text form (lazy_poly.py:467-476 with lazy_text.py:35-71) -- integer-valued
    floats without ``.0``, other floats as ``%g``, coefficients of 1 / -1 implicit, Stream coefficients named a<power>."""
    parts = []
    for power, value in self.terms():
      if hasattr(value, "__iter__"):
        value = ("a%s" % (power,)).replace(".", "_").replace("-", "m")
      parts.append(_term_text(power, value, "x"))
    return _sum_text(parts)

  __repr__ = __str__


def _term_text(power, value, symbol):
  """``value * symbol^power`` as text."""
  if isinstance(value, float):
    value = int(round(value)) if value.is_integer() else "%g" % value
  if power == 0:
    return str(value)
  suffix = "" if power == 1 else "^%s" % (power,)
  if value == 1:
    return "%s%s" % (symbol, suffix)
  if value == -1:
    return "-%s%s" % (symbol, suffix)
  return "%s * %s%s" % (value, symbol, suffix)


def _sum_text(parts):
  """Terms joined by `` + `` / `` - `` (a leading minus of a term becomes the operator); no term: ``0``."""
  if not parts:
    return "0"
  text = parts[0]
  for part in parts[1:]:
    text = "%s - %s" % (text, part[1:]) if part[:1] == "-" else "%s + %s" % (text, part)
  return text


IGNORED_CLASSES.append(Poly)

x = Poly({1: 1})


# ---------------------------------------------------------------------------
# Waring-Lagrange interpolation and the resampler built on it (reference lazy_poly.py:493-603): host arithmetic, one
# interpolation per output sample like the reference (not one of the GPU paths).
# ---------------------------------------------------------------------------
def _make_lagrange():
  import functools
  import operator
  from .strategy import StrategyDict
  sd = StrategyDict("lagrange")

  def func(pairs):
    """The interpolating function of the points ``(x, y)`` in ``pairs``: ``f(k) = sum_j y_j prod_{r != j} (k - x_r) / (x_j - x_r)``,
    the factors multiplied and the terms added left to right as the reference does (lazy_poly.py:496-516)."""
    xv, yv = zip(*pairs)

    def interpolator(k):
      total = 0
      for j, rj in enumerate(xv):
        weight = functools.reduce(operator.mul, ((k - rk) / (rj - rk) for rk in xv if rj != rk))
        total = total + yv[j] * weight
      return total
    return interpolator

  def poly(pairs):
    """The same interpolator as a Poly in ``x`` (lazy_poly.py:519-536)."""
    return func(pairs)(x)

  sd.strategy("func")(func)
  sd.strategy("poly")(poly)
  return sd


lagrange = _make_lagrange()


def _resample(sig, old=1, new=1, order=3, zero=0.):
  """Generic resampler: every output is the Waring-Lagrange interpolation of the ``order + 1`` neighbouring input samples
  at a position that advances by ``old / new`` per output (a number, or an iterable read once per output); the input is
  thought of as preceded by ``zero`` and is NOT padded at its end (reference lazy_poly.py:538-603).  The stream ends when
  the input does (the reference's generator dies with a RuntimeError there on Python >= 3.7)."""
  from collections import deque
  from .misc import rint
  sig = Stream(sig)
  threshold = .5 * (order + 1)
  step = old / new
  data = deque([zero] * (order + 1), maxlen=order + 1)
  data.extend(sig.take(rint(threshold)))
  idx = int(threshold)
  source = iter(sig)
  steps = iter(step) if hasattr(step, "__iter__") else None
  while True:
    yield lagrange(enumerate(data))(idx)
    if steps is None:
      idx += step
    else:
      try:
        idx += next(steps)
      except StopIteration:
        return
    while idx > threshold:
      try:
        data.append(next(source))
      except StopIteration:
        return
      idx -= 1


def resample(sig, old=1, new=1, order=3, zero=0.):
  return Stream(_resample(sig, old=old, new=new, order=order, zero=zero))


resample.__doc__ = _resample.__doc__
