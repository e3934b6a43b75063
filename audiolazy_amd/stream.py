"""Stream: the lazy iterable the filter protocol returns, and ``blocks``.

Host-side mirror of the parts of the reference's ``Stream`` the filter hot path
touches (reference audiolazy/lazy_stream.py:74-405): iteration, elementwise
operators, ``take`` / ``peek`` / ``skip`` / ``limit`` / ``append`` / ``map`` /
``copy`` and ``blocks`` (the feed of the blocked engine, lazy_stream.py:215-220
-> lazy_misc.py:74-129).  Two deliberate differences from the reference under
Python >= 3.7: ``take(n)`` / ``limit(n)`` / ``peek(n)`` on a stream that ends
early return what was there (the reference dies with RuntimeError because of
PEP 479, lazy_stream.py:292, 348).
"""
import collections
import itertools
import operator


def blocks(seq, size=None, hop=None, padval=0.):
  """Blockenizer with the reference's semantics (lazy_misc.py:74-129).

  Yields consecutive windows of ``size`` items, each starting ``hop`` items
  after the previous one (default hop = size); ``hop < size`` overlaps,
  ``hop > size`` skips items; a trailing partial window is padded with
  ``padval``.  Like the reference, the SAME deque object is yielded every
  time: copy it if you keep it.
  """
  if hop is None:
    hop = size
  win = collections.deque(maxlen=size)
  fill = 0        # items of the current window already seen (negative: still skipping)
  for item in seq:
    if fill < 0:  # hop > size: drop the items between two windows
      fill += 1
      continue
    win.append(item)
    fill += 1
    if fill == size:
      yield win
      fill = size - hop
  if fill > max(size - hop, 0):
    for _ in range(size - fill):
      win.append(padval)
    yield win


# classes whose own reflected operators must win over the elementwise Stream ones: filters and
# polynomials (``gain_stream * z ** -1`` is a time-varying filter, not a Stream of filters;
# reference lazy_stream.py:47-51 ``__ignored_classes__``).  Filled in by poly.py / filters.py.
IGNORED_CLASSES = []


def _binary(op):
  def method(self, other):
    if isinstance(other, tuple(IGNORED_CLASSES)):
      return NotImplemented
    if isinstance(other, Stream) or (hasattr(other, "__iter__") and not hasattr(other, "__len__")):
      return Stream(map(op, iter(self), iter(other)))
    return Stream(op(item, other) for item in self)
  return method


def _rbinary(op):
  def method(self, other):
    if isinstance(other, tuple(IGNORED_CLASSES)):
      return NotImplemented
    return Stream(op(other, item) for item in self)
  return method


class Stream(object):
  """Lazy, single-pass iterable with elementwise operators.

  ``Stream(iterable)`` wraps it; ``Stream(a, b, c)`` (or a single non-iterable)
  is the finite stream of those items, like the reference's constructor
  (lazy_stream.py:150-178).
  """

  def __init__(self, *items):
    if len(items) == 1 and hasattr(items[0], "__iter__"):
      self._it = iter(items[0])
    else:
      self._it = iter(items)

  def __iter__(self):
    return self._it

  def __next__(self):
    return next(self._it)

  next = __next__

  # -- consumption -----------------------------------------------------------
  def take(self, n=None, constructor=list):
    """Next item (n is None) or a ``constructor`` of the next n items."""
    if n is None:
      return next(self._it)
    if n == float("inf"):
      return constructor(self._it)
    return constructor(itertools.islice(self._it, int(n)))

  def peek(self, n=None, constructor=list):
    """Like take, without consuming."""
    if n is None:
      first = next(self._it)
      self._it = itertools.chain([first], self._it)
      return first
    head = list(itertools.islice(self._it, int(n)))
    self._it = itertools.chain(head, self._it)
    return constructor(head)

  def skip(self, n):
    for _ in itertools.islice(self._it, int(n)):
      pass
    return self

  def limit(self, n):
    self._it = itertools.islice(self._it, int(n))
    return self

  def append(self, *others):
    self._it = itertools.chain(self._it, *[iter(o) if hasattr(o, "__iter__") else [o] for o in others])
    return self

  def map(self, func):
    self._it = map(func, self._it)
    return self

  def filter(self, func):
    self._it = filter(func, self._it)
    return self

  def copy(self):
    """Independent copy (itertools.tee underneath, like the reference :235-240)."""
    self._it, other = itertools.tee(self._it, 2)
    return Stream(other)

  def blocks(self, *args, **kwargs):
    """Stream of blocks; see :func:`blocks` (reference lazy_stream.py:215-220)."""
    return Stream(blocks(iter(self), *args, **kwargs))

  # -- elementwise operators ---------------------------------------------------
  __add__, __radd__ = _binary(operator.add), _rbinary(operator.add)
  __sub__, __rsub__ = _binary(operator.sub), _rbinary(operator.sub)
  __mul__, __rmul__ = _binary(operator.mul), _rbinary(operator.mul)
  __truediv__, __rtruediv__ = _binary(operator.truediv), _rbinary(operator.truediv)
  __pow__, __rpow__ = _binary(operator.pow), _rbinary(operator.pow)

  def __neg__(self):
    return Stream(-item for item in self)

  def __abs__(self):
    return Stream(abs(item) for item in self)
