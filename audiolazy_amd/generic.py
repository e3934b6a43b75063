"""The per-sample path for items the float64 engine does not take.

The reference's ``LinearFilter.__call__`` (audiolazy/lazy_filters.py:141-264) is type-generic: whatever
supports ``*``, ``+``, unary ``-`` and ``/`` can be a sample -- Python ints stay ints when everything is an int
(doctest :735-742: ``[4, 10, 11, 0, 2]``), complex numbers, NumPy matrices with matrix-valued coefficient
Streams (tests/test_filters_extdep.py:49-89), SymPy symbols.  The GPU engine computes in float64 and takes
real scalars and rows of them; everything else runs here, one sample per ``next`` like the reference, with the
reference's semantics restated:

* Direct Form I, one sum per sample, terms left to right: numerator terms by ascending delay, then denominator
  terms by ascending delay (:197-224);
* a constant coefficient that equals 0 contributes no term, 1 contributes the bare item, -1 its negation
  (:205-210, :219-224); a denominator term is ``-a_k * m_k`` with the coefficient negated FIRST (matrices do
  not commute); a coefficient that is an iterable contributes ``next(b_k) * d_k`` / ``-next(a_k) * m_k``;
* ``(sum) / a0`` unless a0 is 1, ``-(sum)`` when it is -1 (:233-237); an a0 that is a series is divided out of
  every other coefficient sample by sample first (:166-174); a filter without terms yields ``zero`` per item;
* ``memory``: the FIRST ``len(den) - 1`` items are y[-1], y[-2], ...; a short one is LEFT-padded with ``zero``
  (:185-195); a callable is called with the size; ``zero`` is every past input.

* constants reach the reference's generated source as TEXT (``"{value} * d{idx}".format(...)``, :209, :224; the gain
  :236; ``zero`` of a filter without terms :229-231), so an object that is not its own literal is re-read by the
  Python parser: ``Fraction(3, 5)`` becomes ``3/5 * d0`` -- float arithmetic --, a ``Fraction`` gain ``a/b`` makes
  ``(...) / a/b`` (divide by a, then by b), ``-1j`` comes back as ``(-0-1j)`` with a POSITIVE zero real part,
  ``numpy.float64`` / ``Decimal`` as plain floats, a ``zero`` of ``Fraction(3)`` as the int 3.  Restated here by
  compiling the same fragments (``_fragment``); plain ``float`` / ``int`` constants are their own literals
  (``repr`` round-trips) and are used as they are.

This is product code with no GPU in it (slow by nature: it exists so that a pipeline written for the reference
keeps working when such items reach a filter); it never imports ``oracle/``.
"""
import itertools
import numbers

import numpy as np

_REAL_SCALARS = (float, int, np.floating, np.integer)


def is_series(v):
  return hasattr(v, "__iter__")


def is_engine_scalar(v):
  """A real number the float64 engine represents exactly enough: Python / NumPy floats and ints."""
  return isinstance(v, _REAL_SCALARS) and not isinstance(v, complex)


def is_engine_item(item):
  """An input item the engine takes: a real scalar, or a 1-D row of real scalars (C parallel streams)."""
  if is_engine_scalar(item):
    return True
  if isinstance(item, np.matrix) or not hasattr(item, "__len__"):
    return False
  try:
    arr = np.asarray(item)
  except Exception:
    return False
  return arr.ndim == 1 and arr.dtype.kind in "fiub"


def all_int_configuration(numlist, denlist, memory, zero):
  """True when no float can enter the arithmetic except through the items: integer coefficients, and integer
  ``zero`` / ``memory`` wherever the filter reads them.  The reference then keeps integer items integers
  (:735-742), so such a call is not the float engine's."""
  is_int = lambda v: isinstance(v, (int, np.integer))      # (bools included: True * 3 is the int 3, as in the reference)
  present = lambda c: is_series(c) or not is_engine_scalar(c) or c != 0   # (a zero coefficient contributes no term)
  b, a = list(numlist), list(denlist)
  coefs = [c for c in b + a if not is_series(c) and present(c)]
  if not coefs or not all(is_int(c) for c in coefs):
    return False
  lm = len(a) - 1
  short_memory = lm > 0 and (memory is None or callable(memory))
  if not (memory is None or callable(memory)):
    if not hasattr(memory, "__len__"):
      return False       # a one-shot iterator is never pulled here: callers stage it first (read_memory)
    mem = list(memory)[:lm]
    if not all(is_int(v) for v in mem):
      return False
    short_memory = len(mem) < lm
  # ``zero`` is read as a past input by numerator taps at delay >= 1 and fills a missing / short memory
  zero_read = any(present(c) for c in b[1:]) or short_memory
  return is_int(zero) or not zero_read


def coefficients_fit_engine(numlist, denlist):
  """Constant coefficients must be real scalars for the engine (series are looked at when they are pulled)."""
  return all(is_series(c) or is_engine_scalar(c) for c in list(numlist) + list(denlist))


def read_memory(memory, size, zero):
  """The ``memory`` argument as the reference reads it AT CALL TIME (:185-195): None stays None (all ``zero``);
  a callable is called with the size; from an iterable the first ``size`` items are taken -- by the reference's
  own ``takewhile`` over ``enumerate``, which also draws the item that ends it from a one-shot iterator -- and a
  short one is LEFT-padded with ``zero``.  Returns a list of exactly ``size`` items: staged once, it can be
  handed to the gate, the engine and the per-sample path alike without anything being pulled twice."""
  if memory is None:
    return None
  if not is_series(memory):
    memory = memory(size)
  got = [item for _unused, item in itertools.takewhile(lambda pair: pair[0] < size, enumerate(memory))]
  return [zero] * (size - len(got)) + got


def initial_memory(memory, size, zero):
  """``memory`` argument -> [y[-1], y[-2], ...] of exactly ``size`` items (reference :185-195)."""
  if memory is None:
    return [zero] * size
  if not is_series(memory):
    memory = memory(size)
  got = list(itertools.islice(memory, size))
  return [zero] * (size - len(got)) + got


def _own_literal(v):
  """Plain Python floats and ints (not bools, not subclasses) format to a literal of themselves."""
  return type(v) in (float, int) and v == v and v not in (float("inf"), float("-inf"))


def _fragment(template, value):
  """``lambda _v: <the reference's source fragment with the constant formatted in>``, compiled in an empty namespace
  like the reference's generated function (lazy_filters.py:98-106): whatever the text means to the parser is what the
  filter computes.  A constant whose text is not an expression raises here what the reference raises when it
  defines its generator (SyntaxError / NameError at call time)."""
  return eval("lambda _v: " + template.format(value=value), {})     # (names resolve when the fragment first runs: the first next(), as in the reference)


def df1(numlist, denlist, seq, memory=None, zero=0.):
  """Generator of output items: the difference equation evaluated in Python arithmetic on whatever the items
  are.  ``numlist`` / ``denlist``: coefficients by delay, constants or iterables (one value per output item)."""
  b, a = list(numlist), list(denlist)
  if not a:
    raise ZeroDivisionError("Invalid filter gain")
  gain = a[0]
  if is_series(gain):
    # a0[n] varies: every other coefficient becomes c_k[n] * (1 / a0[n]) and a0 is 1 (:166-174)
    nseries = sum(1 for c in b) + sum(1 for c in a[1:])
    invs = itertools.tee((1 / g for g in gain), max(nseries, 1))
    scaled = lambda c, inv: (ck * iv for ck, iv in zip(c if is_series(c) else itertools.repeat(c), inv))
    b = [scaled(c, invs[i]) for i, c in enumerate(b)]
    a = [1] + [scaled(c, invs[len(b) + i]) for i, c in enumerate(a[1:])]
    gain = 1
  elif gain == 0:
    raise ZeroDivisionError("Invalid filter gain")
  lb, lm = len(b), len(a) - 1
  mem = initial_memory(memory, lm, zero)
  # the terms of the sum, in the reference's order
  plan = []
  for k, c in enumerate(b):
    if is_series(c):
      plan.append(("x*s", k, iter(c)))
    elif c == 1:
      plan.append(("x", k, None))
    elif c == -1:
      plan.append(("-x", k, None))
    elif c != 0:
      plan.append(("x*c", k, c) if _own_literal(c) else ("x*t", k, _fragment("{value} * _v", c)))
  for k, c in enumerate(a):
    if k == 0:
      continue
    if is_series(c):
      plan.append(("y*s", k, iter(c)))
    elif c == -1:
      plan.append(("y", k, None))
    elif c == 1:
      plan.append(("-y", k, None))
    elif c != 0:
      plan.append(("y*c", k, -c) if _own_literal(c) else ("y*t", k, _fragment("-{value} * _v", c)))
  if not plan:
    if not _own_literal(zero):
      zero = eval("{zero}".format(zero=zero), {})          # ``yield {zero}`` (:229-231)
  elif gain != -1 and gain != 1 and not _own_literal(gain):
    gain = _fragment("(_v) / {value}", gain)               # ``({expr}) / {gain}`` (:236)
  return _run(plan, gain, iter(seq), mem, zero, lb)


def _run(plan, gain, items, mem, zero, lb):
  if not plan:
    for _unused in items:
      yield zero
    return
  past_in = [zero] * (lb - 1)        # d1, d2, ...
  past_out = list(mem)               # m1, m2, ...
  for item in items:
    total = None
    try:
      for kind, k, arg in plan:
        if kind[0] == "x" or kind[1:2] == "x":
          src = item if k == 0 else past_in[k - 1]
        else:
          src = past_out[k - 1]
        if kind in ("x", "y"):
          term = src
        elif kind in ("-x", "-y"):
          term = -src
        elif kind in ("x*c", "y*c"):
          term = arg * src
        elif kind in ("x*t", "y*t"):
          term = arg(src)
        elif kind == "x*s":
          term = next(arg) * src
        else:
          term = -next(arg) * src
        total = term if total is None else total + term
    except StopIteration:            # a coefficient series that ends, ends the output (the reference's generator dies there)
      return
    if callable(gain):
      total = gain(total)
    elif gain == -1:
      total = -(total)
    elif gain != 1:
      total = (total) / gain
    yield total
    if past_out:
      past_out.insert(0, total)
      past_out.pop()
    if past_in:
      past_in.insert(0, item)
      past_in.pop()
