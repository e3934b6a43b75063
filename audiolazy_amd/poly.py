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
  __slots__ = ("_t",)

  def __init__(self, data=None):
    if data is None:
      terms = {}
    elif isinstance(data, Poly):
      terms = dict(data._t)
    elif isinstance(data, dict):
      terms = dict(data)
    elif isinstance(data, (list, tuple)):
      terms = dict(enumerate(data))
    else:
      terms = {0: data}
    clean = {}
    for power, coef in terms.items():
      if isinstance(power, float) and power.is_integer():
        power = int(power)
      if isinstance(coef, Stream) or coef != 0:     # a Stream is never "zero" (lazy_poly.py:138)
        clean[power] = coef
    self._t = clean

  # -- views -----------------------------------------------------------------
  def terms(self, reverse=False):
    """(power, coefficient) pairs sorted by power (lazy_poly.py:170-199)."""
    for k in sorted(self._t, reverse=reverse):
      yield k, self._t[k]

  def values(self):
    """Dense coefficient list for powers 0 .. order (lazy_poly.py:159-168)."""
    if not self._t:
      return []
    return [self._t.get(k, 0.) for k in range(self.order + 1)]

  @property
  def order(self):
    if any((not isinstance(k, int)) or k < 0 for k in self._t):
      raise AttributeError("Power needs to be positive integers")
    return max(self._t) if self._t else 0

  def is_polynomial(self):
    return all(isinstance(k, int) and k >= 0 for k in self._t)

  def __len__(self):
    return len(self._t)

  def __getitem__(self, power):
    return self._t.get(power, 0.)

  def __eq__(self, other):
    if not isinstance(other, Poly):
      other = Poly(other)
    if set(self._t) != set(other._t):
      return False
    for k, v in self._t.items():
      w = other._t[k]
      if isinstance(v, Stream) or isinstance(w, Stream):   # Streams compare by identity
        if v is not w:
          return False
      elif v != w:
        return False
    return True

  def __ne__(self, other):
    return not self == other

  def __hash__(self):
    return hash(frozenset(self._t.items()))

  def copy(self):
    """Same terms; Stream coefficients are teed so both polynomials stay usable (:258-262)."""
    return Poly({k: _use(v) for k, v in self._t.items()})

  # -- ring operations ---------------------------------------------------------
  def __neg__(self):
    return Poly({k: -_use(v) for k, v in self._t.items()})

  def __pos__(self):
    return self

  def __add__(self, other):
    if not isinstance(other, Poly):
      other = Poly(other)
    out = {k: _use(v) for k, v in self._t.items()}
    for k, v in other._t.items():
      out[k] = (out[k] + _use(v)) if k in out else _use(v)
    return Poly(out)

  __radd__ = lambda self, other: Poly(other) + self

  def __sub__(self, other):
    return self + (-(other if isinstance(other, Poly) else Poly(other)))

  def __rsub__(self, other):
    return Poly(other) + (-self)

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
    return Poly(out)

  def __rmul__(self, other):
    return Poly(other) * self

  def __truediv__(self, other):
    if isinstance(other, Poly):
      if len(other) == 0:
        raise ZeroDivisionError("Dividing Poly instance by zero")
      if len(other) != 1:
        raise NotImplementedError("Can't divide general Poly instances")
      (shift, value), = other._t.items()
      return Poly({k - shift: _use(v) / _use(value) for k, v in self._t.items()})
    return Poly({k: _use(v) / _use(other) for k, v in self._t.items()})

  def __pow__(self, n):
    if isinstance(n, Poly):
      if any(k != 0 for k in n._t):
        raise NotImplementedError("Can't power general Poly instances")
      n = n[0]
    if n == 0:
      return Poly(1)
    if len(self._t) == 0:
      return Poly()
    if len(self._t) == 1:
      (k, v), = self._t.items()
      if isinstance(v, Stream):
        return Poly({k * n: _use(v) ** n})
      return Poly({k * n: 1 if v == 1 else v ** n})   # lazy_poly.py:445-449
    if not isinstance(n, numbers.Integral) or n < 0:
      raise ValueError("only non-negative integer powers of a multi-term Poly")
    out = self
    for _ in range(n - 1):      # ((p * p) * p) ..., lazy_poly.py:450
      out = out * self
    return out

  # -- calculus / evaluation -----------------------------------------------------
  def diff(self, n=1):
    """n-th derivative (lazy_poly.py:259-266)."""
    terms = self._t
    for _ in range(n):
      terms = {k - 1: k * v for k, v in terms.items() if k != 0}
    return Poly(terms)

  def __call__(self, value):
    """Evaluate at a number or a Stream (elementwise), or substitute another Poly / algebraic
    object.

    Substitution is ``sum(coeff * value ** power)`` over the terms in insertion order
    (lazy_poly.py:313-316); numbers and Streams use the Horner-like scheme for plain polynomials
    -- merged steps for missing powers -- and the direct sum otherwise (:318-349).  A Stream is
    teed once per use (``thub``), so it may be any single-pass iterable.
    """
    from .stream import thub
    # (hasattr is useless on a Stream: attribute access is elementwise there)
    if isinstance(value, Poly) or (not isinstance(value, Stream) and hasattr(value, "numpoly")):
      total = 0
      for power, coef in self._t.items():
        total = total + _use(coef) * value ** power
      return Poly(total) if isinstance(value, Poly) else total
    if not self._t:
      return 0.
    if not isinstance(value, Stream) and not hasattr(value, "__iter__"):
      if value is None:
        raise TypeError("cannot evaluate a non-empty Poly at None")
      if value == 0:
        return self[0]
    value = thub(value, len(self._t))
    if self.is_polynomial():
      pairs = [(k, _use(v)) for k, v in self.terms(reverse=True)]
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

  def __repr__(self):
    if not self._t:
      return "0"
    return " + ".join("%r * x^%r" % (v, k) for k, v in self.terms())


IGNORED_CLASSES.append(Poly)

x = Poly({1: 1})
