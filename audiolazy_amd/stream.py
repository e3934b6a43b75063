"""Stream: the lazy iterable the filter protocol returns, ``blocks`` and ``thub``.

Host-side mirror of the reference's ``Stream`` (audiolazy/lazy_stream.py:74-405): the
constructor rules (one iterable: that iterable; one non-iterable: endless repeat; several
non-iterables: endless cycle; several iterables: chained), elementwise operators of every
kind (arithmetic, bitwise, comparisons; reflected; unary), ``take`` / ``peek`` / ``skip`` /
``limit`` / ``append`` / ``map`` / ``filter`` / ``copy`` / elementwise attribute access and
calls, and ``blocks`` (the feed of the blocked engine, lazy_stream.py:215-220 ->
lazy_misc.py:74-129); ``thub`` is the tee hub the reference uses to let one Stream appear
several times in an expression (:469-630).  One deliberate difference under Python >= 3.7:
``take(n)`` / ``limit(n)`` / ``peek(n)`` on a stream that ends early return what was there
(the reference dies with RuntimeError because of PEP 479, lazy_stream.py:292, 348).
"""
import collections
import collections.abc
import itertools
import numbers
import operator


def rint(x, step=1):
  """Nearest multiple of ``step`` (an integer), halves away from zero -- the rounding the reference
  applies to durations and ``take`` counts (lazy_misc.py:44-71); unlike the builtin ``round``
  (halves to even)."""
  import math
  q = x / step
  n = int(math.floor(q + .5)) if q >= 0 else -int(math.floor(-q + .5))
  return n * step


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


def _is_iterable(obj):
  return isinstance(obj, collections.abc.Iterable)    # (a class that merely defines __iter__ is not)


def _binary(op):
  def method(self, other):
    if isinstance(other, tuple(IGNORED_CLASSES)):
      return NotImplemented
    if _is_iterable(other):
      return Stream(map(op, iter(self), iter(other)))
    return Stream(map(lambda a: op(a, other), iter(self)))
  return method


def _rbinary(op):
  def method(self, other):
    if isinstance(other, tuple(IGNORED_CLASSES)):
      return NotImplemented
    if _is_iterable(other):
      return Stream(map(op, iter(other), iter(self)))
    return Stream(map(lambda a: op(other, a), iter(self)))
  return method


def _unary(op):
  def method(self):
    return Stream(map(op, iter(self)))
  return method


class Stream(object):
  """Lazy, single-pass iterable with elementwise operators (reference lazy_stream.py:74-405).

  ``Stream(iterable)`` wraps it; ``Stream(5)`` repeats 5 for ever; ``Stream(1, 2, 3)`` cycles
  through the items for ever; ``Stream(it1, it2)`` chains iterables (:137-191).
  """

  def __init__(self, *dargs):
    if len(dargs) == 0:
      raise TypeError("Missing argument(s)")
    if len(dargs) == 1:
      self._data = iter(dargs[0]) if _is_iterable(dargs[0]) else itertools.repeat(dargs[0])
    elif all(_is_iterable(arg) for arg in dargs):
      self._data = itertools.chain(*dargs)
    elif not any(_is_iterable(arg) for arg in dargs):
      self._data = itertools.cycle(dargs)
    else:
      raise TypeError("Input with both iterables and non-iterables")

  def __iter__(self):
    return self._data

  def __bool__(self):
    raise TypeError("Streams can't be used as booleans; freeze the stream first with "
                    "list(my_stream) or tuple(my_stream), or use the bitwise operators")

  # -- consumption -----------------------------------------------------------
  def take(self, n=None, constructor=list):
    """Next item (n is None: StopIteration when there is none) or a ``constructor`` of the
    next n items (fewer when the stream ends); floats are rounded, ``inf`` takes all."""
    data = iter(self)
    if n is None:
      return next(data)
    if isinstance(n, float):
      if n == float("inf"):
        return constructor(data)
      n = rint(n) if n > 0 else 0           # so that -inf and nan take nothing
    elif not isinstance(n, numbers.Integral):
      raise TypeError("take / peek need a number of items, not %r" % (type(n).__name__,))
    return constructor(itertools.islice(data, max(int(n), 0)))

  def copy(self):
    """Independent copy; this stream stays usable (itertools.tee underneath, :294-301)."""
    self._data, other = itertools.tee(self._data, 2)
    return Stream(other)

  tee = copy

  def peek(self, n=None, constructor=list):
    """Like take, without consuming (:303-322)."""
    return self.copy().take(n=n, constructor=constructor)

  def skip(self, n):
    """Throw away the first n items, lazily (:324-341)."""
    def skipper(data, count):
      for _ in itertools.islice(data, count):
        pass
      for item in data:
        yield item
    self._data = skipper(self._data, int(round(n)))
    return self

  def limit(self, n):
    """End the stream after n items (:343-349)."""
    self._data = itertools.islice(self._data, int(round(n)))
    return self

  def append(self, *other):
    """Chain other stream(s) / items after this one: ``Stream(self, *other)`` (:366-374)."""
    self._data = itertools.chain(self._data, iter(Stream(*other)))
    return self

  def map(self, func):
    self._data = map(func, self._data)
    return self

  def filter(self, func):
    self._data = filter(func, self._data)
    return self

  def blocks(self, *args, **kwargs):
    """Stream of blocks; see :func:`blocks` (reference lazy_stream.py:215-220)."""
    return Stream(blocks(iter(self), *args, **kwargs))

  def __getattr__(self, name):
    """Elementwise attribute access, e.g. ``stream.real`` (:351-357)."""
    if name.startswith("__") or name in ("next", "_data"):
      raise AttributeError(name)       # Streams are iterable, not iterators
    return Stream(getattr(item, name) for item in iter(self))

  def __call__(self, *args, **kwargs):
    """Elementwise call of a stream of callables (:359-364)."""
    return Stream(item(*args, **kwargs) for item in iter(self))

  def __abs__(self):
    return self.map(abs)

  # -- elementwise operators ---------------------------------------------------
  __add__, __radd__ = _binary(operator.add), _rbinary(operator.add)
  __sub__, __rsub__ = _binary(operator.sub), _rbinary(operator.sub)
  __mul__, __rmul__ = _binary(operator.mul), _rbinary(operator.mul)
  __truediv__, __rtruediv__ = _binary(operator.truediv), _rbinary(operator.truediv)
  __floordiv__, __rfloordiv__ = _binary(operator.floordiv), _rbinary(operator.floordiv)
  __mod__, __rmod__ = _binary(operator.mod), _rbinary(operator.mod)
  __pow__, __rpow__ = _binary(operator.pow), _rbinary(operator.pow)
  __lshift__, __rlshift__ = _binary(operator.lshift), _rbinary(operator.lshift)
  __rshift__, __rrshift__ = _binary(operator.rshift), _rbinary(operator.rshift)
  __and__, __rand__ = _binary(operator.and_), _rbinary(operator.and_)
  __or__, __ror__ = _binary(operator.or_), _rbinary(operator.or_)
  __xor__, __rxor__ = _binary(operator.xor), _rbinary(operator.xor)
  __lt__, __le__ = _binary(operator.lt), _binary(operator.le)
  __gt__, __ge__ = _binary(operator.gt), _binary(operator.ge)
  __eq__, __ne__ = _binary(operator.eq), _binary(operator.ne)
  __hash__ = object.__hash__
  __neg__, __pos__, __invert__ = _unary(operator.neg), _unary(operator.pos), _unary(operator.invert)


class ControlStream(Stream):
  """Endless Stream of a control value that can be changed at any time through ``.value``
  (reference lazy_stream.py:436-462).  Note that the GPU engine pulls its inputs and coefficient
  streams a block at a time: a change is heard one block later (see ``block_size``)."""

  def __init__(self, value):
    self.value = value

    def data_generator():
      while True:
        yield self.value
    super(ControlStream, self).__init__(data_generator())


def _mix_starts(deltas):
  """Output sample at which each added track starts, by Streamix's clock: a counter that starts
  at 0.5, gains 1 per output sample and loses the track's delta when the track starts; a track
  starts as soon as the counter has reached its delta (reference lazy_stream.py:689-697)."""
  import math
  starts, count, n = [], 0.5, 0
  for delta in deltas:
    if delta < 0:
      raise ValueError("Delta time should be always positive")
    wait = max(0, int(math.ceil(delta - count)))
    n += wait
    count = count + float(wait) - delta
    starts.append(n)
  return starts


class _MixSchedule(object):
  """State of one Streamix: the tracks not yet started (in the order added, each with its delta),
  the voices sounding now, and the mixer's clock.  The clock is a counter that starts at 0.5,
  gains 1 per output sample and loses a track's delta when that track starts; a track starts as
  soon as the clock has reached its delta (`_mix_starts` is the closed form of this rule)."""
  _END = object()

  def __init__(self, zero):
    self.pending = collections.deque()
    self.voices = []
    self.clock = 0.5
    self.zero = zero

  def admit_due(self):
    while self.pending and self.pending[0][0] <= self.clock:
      delta, voice = self.pending.popleft()
      self.clock -= delta
      self.voices.append(voice)

  def sum_voices(self):
    """One sample of every sounding voice added to ``zero`` in the order the voices were added;
    voices that have ended are dropped (they contribute nothing to this sample)."""
    total, alive = self.zero, []
    for voice in self.voices:
      item = next(voice, self._END)
      if item is not self._END:
        total += item
        alive.append(voice)
    self.voices = alive
    return total


def avoid_stream(cls):
  """Class decorator: instances are left alone by a Stream's operators -- the other operand's own reflected operator
  decides (lazy_stream.py:400-414)."""
  IGNORED_CLASSES.append(cls)
  return cls


def tostream(func, module_name=None):
  """Decorator: the function's result (typically a generator) comes back as a Stream (lazy_stream.py:417-433)."""
  import functools

  @functools.wraps(func)
  def new_func(*args, **kwargs):
    return Stream(func(*args, **kwargs))
  if module_name is not None:
    new_func.__module__ = module_name
  return new_func


class MemoryLeakWarning(Warning):
  """For StreamTeeHub copies that were never used (lazy_stream.py:465-466)."""


class Streamix(Stream):
  """Stream mixer: iterables that enter at their own times, summed sample by sample in the
  order they were added, starting from ``zero`` (reference lazy_stream.py:633-724).
  ``add(delta, data)``: ``delta`` samples (may be float) after the previously added one.
  ``keep=True`` keeps yielding ``zero`` when nothing is left to play.  For tracks that are
  arrays, :func:`audiolazy_amd.bank.mix_tracks` does the same sum on the GPU."""

  def __init__(self, keep=False, zero=0.):
    self.keep = keep
    self._schedule = _MixSchedule(zero)
    super(Streamix, self).__init__(self._samples())

  def _samples(self):
    sched = self._schedule
    while True:
      sched.admit_due()
      sample = sched.sum_voices()
      if not (sched.voices or sched.pending or self.keep):
        return                      # everything has played out (the last sum was of nothing)
      yield sample
      sched.clock += 1.

  def add(self, delta, data):
    if delta < 0:
      raise ValueError("Delta time should be always positive")
    self._schedule.pending.append((delta, iter(data)))


class StreamTeeHub(Stream):
  """A Stream that hands out up to ``n`` independent copies of itself, one per use (every
  ``iter()`` -- hence every operator, filter call, ``limit`` / ``skip`` / ``append`` / ``map`` /
  ``filter`` / ``blocks``, each of which returns a plain Stream of that copy), so that one signal
  can appear several times in an expression.  ``peek`` and ``copy`` do not use a copy up;
  ``take`` is refused (cast to Stream first).  Reference lazy_stream.py:469-571."""

  def __init__(self, data, n):
    self._iters = list(itertools.tee(iter(data), n))

  def __iter__(self):
    try:
      return self._iters.pop()
    except IndexError:
      raise IndexError("StreamTeeHub has no more copies left to use.")

  def take(self, *args, **kwargs):
    raise AttributeError("Use peek or cast to Stream.")

  def copy(self):
    if not self._iters:
      iter(self)            # raises the usual IndexError
    first, other = itertools.tee(self._iters[0], 2)
    self._iters[0] = first
    return Stream(other)

  def peek(self, n=None, constructor=list):
    return self.copy().take(n=n, constructor=constructor)

  def limit(self, n):
    return Stream(self).limit(n)

  def skip(self, n):
    return Stream(self).skip(n)

  def append(self, *other):
    return Stream(self).append(*other)

  def map(self, func):
    return Stream(self).map(func)

  def filter(self, func):
    return Stream(self).filter(func)

  def blocks(self, *args, **kwargs):
    return Stream(self).blocks(*args, **kwargs)

  def __abs__(self):
    return Stream(self).map(abs)

  def __getattr__(self, name):
    if name.startswith("__") or name in ("next", "_iters"):
      raise AttributeError(name)
    return Stream(getattr(item, name) for item in iter(self))


def thub(data, n):
  """Tee hub: ``data`` usable ``n`` times in what follows.  Non-iterables come back unchanged,
  so designs can be written once for numbers and Streams (reference lazy_stream.py:573-630)."""
  return StreamTeeHub(data, n) if _is_iterable(data) else data


# the itertools names filter expressions are written with, returning Streams like the
# reference's wrappers (audiolazy/lazy_itertools.py:40-60)
def cycle(iterable):
  return Stream(itertools.cycle(iterable))


def repeat(item, times=None):
  return Stream(itertools.repeat(item) if times is None else itertools.repeat(item, times))


def count(start=0, step=1):
  return Stream(itertools.count(start, step))


def chain(*iterables):
  return Stream(itertools.chain(*iterables))


def zero_pad(seq, left=0, right=0, zero=0.):
  """``left`` zeros, the sequence, ``right`` zeros (reference lazy_misc.py:132-160)."""
  def gen():
    for _ in range(left):
      yield zero
    for item in seq:
      yield item
    for _ in range(right):
      yield zero
  return gen()
