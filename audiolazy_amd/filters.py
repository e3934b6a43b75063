"""ZFilter / z algebra, filter containers and the 1st/2nd-order designs.

Host-side mirror of the reference's operator surface for the hot path
(reference audiolazy/lazy_filters.py):

  LinearFilter / ZFilter / z     :110-892   (+ - * / ** diff, numlist/denlist, call)
  CascadeFilter, ParallelFilter  :970-1084
  comb, resonator                :1087-1310
  lowpass, highpass              :1313-1495

Design and algebra run on the host in float64 and reproduce the reference's
coefficients bit for bit (tests/test_designs_golden.py).  *Calling* a filter
executes on the GPU through :class:`audiolazy_amd.bank.FilterBank`; there is no
per-sample Python execution path in this package.
"""
import cmath
import math
import numbers
import operator
from functools import reduce as _reduce

from .poly import Poly
from .strategy import StrategyDict

__all__ = ["LinearFilter", "ZFilter", "z", "CascadeFilter", "ParallelFilter", "comb", "resonator",
           "lowpass", "highpass"]


class LinearFilterProperties(object):
  """Coefficient views shared by every linear filter -- ZFilter, CascadeFilter, ParallelFilter -- from its
  ``numpoly`` / ``denpoly`` (reference lazy_filters.py:47-95)."""

  @staticmethod
  def _causal_list(poly):
    if any(k < 0 for k, _ in poly.terms()):
      raise ValueError("Non-causal filter")
    return poly.values()

  @property
  def numlist(self):
    return self._causal_list(self.numpoly)

  @property
  def denlist(self):
    return self._causal_list(self.denpoly)

  numerator, denominator = numlist, denlist

  @property
  def numdict(self):
    return dict(self.numpoly.terms())

  @property
  def dendict(self):
    return dict(self.denpoly.terms())

  @property
  def numpolyz(self):
    """The numerator as a Poly in ``z`` instead of ``z ** -1`` (for roots; reference :77-85)."""
    return Poly(self.numerator[::-1])

  @property
  def denpolyz(self):
    return Poly(self.denominator[::-1])


class LinearFilter(LinearFilterProperties):
  """A rational transfer function in ``z ** -1`` (reference lazy_filters.py:110-338).

  ``numpoly`` / ``denpoly`` are :class:`Poly` in x = z ** -1; ``numlist`` /
  ``denlist`` are the dense coefficient lists the engine consumes.
  """

  def __init__(self, numerator=None, denominator=None):
    if isinstance(numerator, LinearFilter):          # filter type cast (reference :115-120)
      if denominator is not None:
        numerator = numerator / denominator
      self.numpoly, self.denpoly = numerator.numpoly, numerator.denpoly
    else:
      self.numpoly = Poly(numerator)
      self.denpoly = Poly({0: 1} if denominator is None else denominator)
    # the denominator starts at z ** 0: a common delay / advance factor is cancelled (:126-132); a
    # denominator with no terms (division by the zero filter) is the reference's ``min()`` of nothing
    power = min(k for k, _ in self.denpoly.terms())   # ValueError when there is no term, like the reference
    if power != 0:
      shift = Poly([0, 1]) ** -power
      self.numpoly = self.numpoly * shift
      self.denpoly = self.denpoly * shift

  def __iter__(self):
    """(numdict, dendict): what the reference's helpers compare filters by (reference :134-136)."""
    yield self.numdict
    yield self.dendict

  def is_lti(self):
    """False when some coefficient is a Stream, i.e. varies in time (reference :303-314)."""
    return not any(hasattr(v, "__iter__") for poly in (self.numpoly, self.denpoly) for _, v in poly.terms())

  def is_causal(self):
    return all(k >= 0 for k, _ in self.numpoly.terms())   # reference :316-325

  def copy(self):
    return type(self)(self.numpoly.copy(), self.denpoly.copy())

  def linearize(self):
    """Replace every fractional delay by the linear interpolation of its two integer
    neighbours: ``z ** -4.3`` becomes ``0.7 * z ** -4 + 0.3 * z ** -5`` (reference :339-373);
    this is what makes a ``comb`` at a non-integer period (``karplus_strong``) executable."""
    sides = []
    for poly in (self.numpoly, self.denpoly):
      out = {}
      for k, v in poly.terms():
        if isinstance(k, int) or (isinstance(k, float) and k.is_integer()):
          parts = [(int(k), v)]
        else:
          lo = int(k)
          w_hi = k - lo
          parts = [(lo, v * (1. - w_hi)), (lo + 1, v * w_hi)]
        for key, val in parts:
          out[key] = out[key] + val if key in out else val
      sides.append(out)
    return type(self)(*sides)

  def __eq__(self, other):
    return isinstance(other, LinearFilter) and self.numpoly == other.numpoly \
        and self.denpoly == other.denpoly

  def __ne__(self, other):
    # (the reference's own definition, :683-686: BOTH polynomials must differ, and nothing else is "unequal")
    if isinstance(other, LinearFilter):
      return self.numpoly != other.numpoly and self.denpoly != other.denpoly
    return False

  def __hash__(self):
    return hash((self.numpoly, self.denpoly))

  @property
  def poles(self):
    """Roots of the denominator in ``z`` (reference :640-657; numpy.roots like there)."""
    return self.denpolyz.roots

  @property
  def zeros(self):
    return self.numpolyz.roots

  # -- analysis -----------------------------------------------------------------
  def freq_response(self, freq):
    """H(e^{j freq}), freq in rad/sample (reference :267-301); iterables map elementwise (a
    Stream gives a Stream, other containers their own type)."""
    if hasattr(freq, "__iter__"):
      import types
      from .stream import Stream
      data = (self.freq_response(f) for f in freq)
      if isinstance(freq, types.GeneratorType):
        return data
      return Stream(data) if isinstance(freq, Stream) else type(freq)(data)
    z_ = cmath.exp(-1j * freq)
    den = self.denpoly(z_)
    if den == 0:
      return float("nan")
    return self.numpoly(z_) / den

  # -- execution ------------------------------------------------------------------
  def __call__(self, seq, memory=None, zero=0.):
    """IIR / FIR filtering of any iterable; returns a Stream (reference :141-264).

    Same argument meaning and errors as the reference: ``memory`` holds past
    outputs (first items used, left-padded with ``zero`` when short), ``zero``
    is the value of every past input; non-causal -> ValueError, zero a0 ->
    ZeroDivisionError.  Executed by the GPU engine in blocks.
    """
    if any(k < 0 for k, _ in self.numpoly.terms()) or any(k < 0 for k, _ in self.denpoly.terms()):
      raise ValueError("Non-causal filter")
    from . import generic, timevar
    from .stream import Stream
    numlist, denlist = self.numlist, self.denlist
    series = any(timevar.is_series(v) for v in numlist) or any(timevar.is_series(v) for v in denlist)
    if not series and self.denpoly[0] == 0:
      raise ZeroDivisionError("Invalid filter gain")
    # ``memory`` is read NOW, once, like the reference reads it before it builds its generator (:185-195): a
    # one-shot iterator is drawn from exactly once, and the staged list serves the gate, the engine and the
    # per-sample path alike
    memory = generic.read_memory(memory, len(denlist) - 1, zero)
    # The accept gate: the engine computes in float64 on real scalars and rows of them.  What it
    # cannot represent -- an all-integer configuration (ints stay ints, reference :735-742), complex / matrix /
    # symbolic coefficients or items -- runs on the per-sample path with the reference's semantics.
    if not generic.coefficients_fit_engine(numlist, denlist) or \
       generic.all_int_configuration(numlist, denlist, memory, zero) or not _zero_fits_engine(zero):
      return Stream(generic.df1(numlist, denlist, seq, memory=memory, zero=zero))
    if series:
      # Stream coefficients: the time-varying kernel (reference :197-224) when every value is a real number
      return Stream(_gated(lambda s, cs: timevar.run(cs[0], cs[1], s, memory=memory, zero=zero),
                           lambda s, cs: generic.df1(cs[0], cs[1], s, memory=memory, zero=zero), seq, numlist, denlist))
    from .bank import call_sections
    # (items that are rows of C values are C parallel streams through the same filter -- the
    # reference's vector-valued idiom; call_sections looks at the first item when it is pulled)
    sections = [(numlist or [0.], denlist)]
    return Stream(_gated(lambda s, cs: call_sections(sections, s, memory=memory, zero=zero),
                         lambda s, cs: generic.df1(numlist, denlist, s, memory=memory, zero=zero), seq, numlist, denlist))


def _call_arguments(args, kwargs):
  """(memory, zero, other keyword arguments) of a container call ``filt(seq, *args[1:], **kwargs)``."""
  rest = dict(kwargs)
  memory = args[1] if len(args) > 1 else rest.pop("memory", None)
  zero = args[2] if len(args) > 2 else rest.pop("zero", 0.)
  return memory, zero, rest


def _stage_member_memories(members, memory, zero):
  """``memory`` as the members of a container read it, one after the other AT CALL TIME (the reference's
  ``reduce`` calls every member before anything is iterated, :988-990, :1052-1054): a list serves every member
  from its start, a one-shot iterator is drawn from member by member, a callable is called once per member."""
  from . import generic
  if memory is None:
    return [None] * len(members)
  return [generic.read_memory(memory, len(f.denlist) - 1, zero) for f in members]


def _members_fit_engine(members, memories, zero):
  """Every member an LTI LinearFilter whose call would pass the engine's coefficient / zero / memory gate
  (``memories``: the staged memory of every member)."""
  from . import generic
  if not generic.is_engine_item(zero):
    return False
  for f, memory in zip(members, memories):
    if not generic.coefficients_fit_engine(f.numlist, f.denlist) or \
       generic.all_int_configuration(f.numlist, f.denlist, memory, zero):
      return False
  return True


def _members_are_lti(members):
  return all(isinstance(f, LinearFilter) and f.is_lti() and f.is_causal() and
             all(k >= 0 for k, _ in f.denpoly.terms()) for f in members)


def _zero_fits_engine(zero):
  from . import generic
  return generic.is_engine_item(zero)


def _gated(engine, fallback, seq, numlist, denlist):
  """Iterator behind a filter call: nothing is pulled until the result is iterated (like the reference's
  generator); then the FIRST input item and the first value of every coefficient series decide between the
  GPU engine (real scalars / rows of them) and the per-sample path (everything else).  The peeked values are
  chained back in front of their iterators.  The choice is made inside a one-item generator under
  ``chain.from_iterable``: once made, the items come straight from the chosen iterator (for the engine a chain
  over the lists of whole blocks) -- no Python frame is resumed per sample."""
  import itertools
  from . import generic

  def decide():
    it = iter(seq)
    for first in it:
      break
    else:
      return
    fits = generic.is_engine_item(first)
    sides = []
    for coefs in (numlist, denlist):
      out = []
      for c in coefs:
        if generic.is_series(c):
          ci = iter(c)
          for head in ci:
            fits = fits and generic.is_engine_item(head)
            c = itertools.chain([head], ci)
            break
          else:
            c = iter(())
        out.append(c)
      sides.append(out)
    yield iter((engine if fits else fallback)(itertools.chain([first], it), sides))
  return itertools.chain.from_iterable(decide())


class ZFilter(LinearFilter):
  """Linear filter built with Z-transform algebra (reference :710-889).

  ``ZFilter([b0, b1, ...], [a0, a1, ...])`` or expressions on :data:`z`, e.g.
  ``(1 + z ** -1) / (1 - z ** -1)``.
  """

  # -- algebra (reference :744-838) ---------------------------------------------
  @staticmethod
  def _lift(other):
    if isinstance(other, ZFilter):
      return other
    if isinstance(other, LinearFilter):
      raise ValueError("Filter equations have different domains")
    return ZFilter([other])

  def __neg__(self):
    return ZFilter(-self.numpoly, self.denpoly)

  def __pos__(self):
    return self

  def __add__(self, other):
    other = self._lift(other)
    if self.denpoly == other.denpoly:
      return ZFilter(self.numpoly + other.numpoly, self.denpoly)
    return ZFilter(self.numpoly * other.denpoly + other.numpoly * self.denpoly,
                   self.denpoly * other.denpoly)

  def __radd__(self, other):
    return self._lift(other) + self

  def __sub__(self, other):
    return self + (-self._lift(other))

  def __rsub__(self, other):
    return self._lift(other) + (-self)

  def __mul__(self, other):
    if isinstance(other, ZFilter):
      return ZFilter(self.numpoly * other.numpoly, self.denpoly * other.denpoly)
    if isinstance(other, LinearFilter):
      raise ValueError("Filter equations have different domains")
    return ZFilter(self.numpoly * other, self.denpoly)

  def __rmul__(self, other):
    return self._lift(other) * self

  def __truediv__(self, other):
    if isinstance(other, ZFilter):
      return ZFilter(self.numpoly * other.denpoly, self.denpoly * other.numpoly)
    if isinstance(other, LinearFilter):
      raise ValueError("Filter equations have different domains")
    return self * (1 / other)

  def __rtruediv__(self, other):
    return self._lift(other) / self

  def __pow__(self, n):
    if not isinstance(n, numbers.Real):
      raise ValueError("Z-transform powers only valid with integers")
    if n < 0 and (len(self.numpoly) >= 2 or len(self.denpoly) >= 2):
      return ZFilter(self.denpoly, self.numpoly) ** -n
    return ZFilter(self.numpoly ** n, self.denpoly ** n)

  def diff(self, n=1, mul_after=1):
    """n-th derivative with respect to z; every intermediate derivative is
    multiplied by ``mul_after`` before the next one is taken (reference :819-838)."""
    if isinstance(mul_after, ZFilter):
      den = ZFilter(self.denpoly)
      num = ZFilter(self.numpoly)
      for order in range(1, n + 1):
        num = mul_after * (num.diff() * den - order * num * den.diff())
      return num / den ** (n + 1)
    inv = Poly({-1: 1})   # the Poly variable is z ** -1
    den = self.denpoly(inv)
    num = self.numpoly(inv)
    for order in range(1, n + 1):
      num = mul_after * (num.diff() * den - order * num * den.diff())
    return ZFilter(num(inv), self.denpoly ** (n + 1))

  def __call__(self, seq, memory=None, zero=0.):
    """Filter an iterable -- or, given a ZFilter, substitute it for z
    (reference :840-889), e.g. ``filt(1 / z)`` reverses the coefficients."""
    if isinstance(seq, ZFilter):
      # the reference's own expression, term by term (its roundings, its int / float types, its a0)
      return sum(v * seq ** -k for k, v in self.numpoly.terms()) / \
             sum(v * seq ** -k for k, v in self.denpoly.terms())
    return super(ZFilter, self).__call__(seq, memory=memory, zero=zero)

  def __str__(self):
    """The reference's text form (lazy_filters.py:782-817): numerator over denominator in powers of ``z``, the shorter
    line centred, only the numerator when there is no feedback."""
    from .poly import _term_text, _sum_text
    sides = []
    for poly, letter in ((self.numpoly, "b"), (self.denpoly, "a")):
      parts = []
      for power, value in poly.terms():
        if hasattr(value, "__iter__"):
          value = ("%s%s" % (letter, power)).replace(".", "_").replace("-", "m")
        if value != 0.:
          parts.append(_term_text(-power, value, "z"))
      sides.append(parts)
    num = _sum_text(sides[0])
    if not sides[1]:
      raise TypeError("reduce() of empty sequence with no initial value")     # (what the reference's reduce raises)
    den = _sum_text(sides[1])
    if den == "1":
      return num
    line = "-" * max(len(num), len(den))
    pad = " " * (abs(len(num) - len(den)) // 2)
    if len(num) > len(den):
      den = pad + den
    elif len(den) > len(num):
      num = pad + num
    # (lines longer than 80 characters continue below, 80 at a time)
    pieces = [slice(80 * b, 80 * (b + 1)) for b in range(len(line) // 80 + 1)]
    return "\n\n    ...continue...\n\n".join("\n".join([num[p], line[p], den[p]]) for p in pieces)

  __repr__ = __str__


from .stream import IGNORED_CLASSES as _stream_ignored  # noqa: E402
_stream_ignored.append(LinearFilter)

z = ZFilter({-1: 1})   # z ** -1 is the unit delay: the Poly variable is x = z ** -1


# ---------------------------------------------------------------------------
# containers (reference :895-1084)
# ---------------------------------------------------------------------------
def _elementwise_freq(method):
  """``freq`` may be a number or an iterable, mapped elementwise: a Stream gives a Stream, a
  generator a generator, any other container its own type (the reference's ``@elementwise``,
  lazy_misc.py:163-229)."""
  import functools
  import types

  @functools.wraps(method)
  def wrapper(self, freq):
    if hasattr(freq, "__iter__"):
      from .stream import Stream
      data = (method(self, f) for f in freq)
      if isinstance(freq, types.GeneratorType):
        return data
      if isinstance(freq, Stream):
        return Stream(data)
      return type(freq)(data)
    return method(self, freq)
  return wrapper


class FilterList(list, LinearFilterProperties):
  """A list of filters that is itself a filter; a single filter or an iterable of filters
  builds it (reference :906-968).  Items that are not callable (numbers, coefficient lists)
  count as the LinearFilter they cast to (``callables``, :955-961); ``+`` and ``*`` act like on
  lists and keep the class (:893-903)."""

  def __init__(self, *filters):
    if len(filters) == 1 and not callable(filters[0]) and hasattr(filters[0], "__iter__"):
      filters = filters[0]
    super(FilterList, self).__init__(filters)

  @property
  def callables(self):
    return [f if callable(f) else LinearFilter(f) for f in self]

  def is_linear(self):
    return all(isinstance(f, LinearFilter) or (hasattr(f, "is_linear") and f.is_linear())
               for f in self.callables)

  def is_lti(self):
    return self.is_linear() and all(f.is_lti() for f in self.callables)

  def is_causal(self):
    return all(f.is_causal() for f in self.callables if hasattr(f, "is_causal"))

  def __add__(self, other):
    return type(self)(list.__add__(self, other))

  def __radd__(self, other):
    return type(self)(list.__add__(list(other), self))

  def __mul__(self, other):
    return type(self)(list.__mul__(self, other))

  __rmul__ = __mul__

  def __eq__(self, other):
    return type(self) == type(other) and list.__eq__(self, other)

  def __ne__(self, other):
    return type(self) != type(other) or list.__ne__(self, other)

  __hash__ = None

  def _polys(self, name):
    try:
      return [getattr(f, name) for f in self.callables]
    except AttributeError:
      raise AttributeError("Non-linear filter")


class CascadeFilter(FilterList):
  """Filters applied one after the other; the call arguments after the input (``memory`` /
  ``zero``) are forwarded unchanged to every stage (reference :970-1021, call :988-990).  A
  cascade of LTI linear filters runs as ONE fused bank on the GPU; anything else (time-varying
  members, arbitrary callables) is the plain composition of its members."""

  def __call__(self, *args, **kwargs):
    seq = args[0]
    members = self.callables
    if not members:
      from .stream import Stream
      return Stream(seq)
    if _members_are_lti(members) and not any(f.denpoly[0] == 0 for f in members):
      memory, zero, rest = _call_arguments(args, kwargs)
      if set(rest) <= {"block"}:
        # every stage's memory is read now, in stage order, like the reference's reduce does it (:988-990)
        memories = _stage_member_memories(members, memory, zero)
        if _members_fit_engine(members, memories, zero):
          from .bank import call_sections, sections_of
          from .stream import Stream

          def one_by_one(data, _coefs=None):   # items the engine does not take: the members' own gates
            for f, mem in zip(members, memories):
              data = f(data, memory=mem, zero=zero)
            return data
          hists = None if memory is None else memories
          return Stream(_gated(lambda s, cs: call_sections(sections_of(members), s, zero=zero, _hists=hists, **rest),
                               one_by_one, seq, [], []))
        data = seq
        for f, mem in zip(members, memories):
          data = f(data, memory=mem, zero=zero)
        return data
    data = seq
    for f in members:
      data = f(data, *args[1:], **kwargs)
    return data

  @property
  def numpoly(self):
    return _reduce(operator.mul, self._polys("numpoly"))

  @property
  def denpoly(self):
    return _reduce(operator.mul, self._polys("denpoly"))

  @_elementwise_freq
  def freq_response(self, freq):
    members = self.callables
    if any(not hasattr(f, "freq_response") for f in members):
      raise AttributeError("Non-linear filter")
    return _reduce(operator.mul, (f.freq_response(freq) for f in members))

  @property
  def poles(self):
    if not self.is_lti():
      raise AttributeError("Not a LTI filter")
    return _reduce(operator.concat, (f.poles for f in self.callables))

  @property
  def zeros(self):
    if not self.is_lti():
      raise AttributeError("Not a LTI filter")
    return _reduce(operator.concat, (f.zeros for f in self.callables))


class ParallelFilter(FilterList):
  """Filters fed with the same input, outputs summed ``((f1 + f2) + f3) ...``
  (reference :1024-1084, call :1048-1054).  LTI linear members run as one OUTER bank on the GPU
  followed by the ordered sum on the device; otherwise every member gets its own copy of the
  input and the output Streams are added.  An empty list yields ``zero`` per input item."""

  def __call__(self, *args, **kwargs):
    from .stream import Stream
    seq = args[0]
    members = self.callables
    if len(members) == 0:
      zero = kwargs.get("zero", 0.)
      return Stream(zero for _ in seq)
    import itertools

    def one_by_one(src, _coefs=None, memories=None):
      copies = itertools.tee(src, len(members))
      total = None
      for i, (f, part) in enumerate(zip(members, copies)):
        out = f(part, *args[1:], **kwargs) if memories is None else f(part, memory=memories[i], zero=zero)
        out = out if isinstance(out, Stream) else Stream(out)
        total = out if total is None else total + out
      return total
    if _members_are_lti(members) and not any(f.denpoly[0] == 0 for f in members):
      memory, zero, rest = _call_arguments(args, kwargs)
      if set(rest) <= {"block"}:
        # every member's memory is read now, in member order (the reference's reduce, :1052-1054)
        memories = _stage_member_memories(members, memory, zero)
        if _members_fit_engine(members, memories, zero):
          def bank(src, _coefs=None):
            try:
              return self._call_bank(src, memories, zero, rest.get("block"))
            except NotImplementedError:   # coefficients outside the engine's gate
              return one_by_one(src, memories=memories)
          return Stream(_gated(bank, lambda src, _c=None: one_by_one(src, memories=memories), seq, [], []))
        return one_by_one(seq, memories=memories)
    return one_by_one(seq)

  @property
  def numpoly(self):
    if not self.is_linear():
      raise AttributeError("Non-linear filter")
    return _reduce(operator.add, self).numpoly        # (the members as they are, like the reference :1056-1060)

  @property
  def denpoly(self):
    return _reduce(operator.mul, self._polys("denpoly"))

  @_elementwise_freq
  def freq_response(self, freq):
    members = self.callables
    if any(not hasattr(f, "freq_response") for f in members):
      raise AttributeError("Non-linear filter")
    return _reduce(operator.add, (f.freq_response(freq) for f in members))

  @property
  def poles(self):
    if not self.is_lti():
      raise AttributeError("Not a LTI filter")
    return _reduce(operator.concat, (f.poles for f in self.callables))

  @property
  def zeros(self):
    if not self.is_lti():
      raise AttributeError("Not a LTI filter")
    return _reduce(operator.add, (ZFilter(f) for f in self)).zeros

  def _call_bank(self, seq, memories, zero, block):
    """All filters as the coefficient sets of one OUTER bank over the single input, then the
    ordered sum over the sets on the device (alz_mix_dev)."""
    import itertools
    import numpy as np
    from .bank import FilterBank, memory_to_hist, sections_of, block_size
    from .stream import Stream
    block = block_size() if block is None else block
    members = self.callables
    for f in members:
      if not f.is_causal():
        raise ValueError("Non-causal filter")
      if f.denpoly[0] == 0:
        raise ZeroDivisionError("Invalid filter gain")
    secs = [sections_of(f)[0] for f in members]
    nb = max(max(len(b) for b, _ in secs), 1)
    na = max(len(a) for _, a in secs)
    b = np.zeros((len(secs), nb))
    a = np.zeros((len(secs), na))
    for i, (bi, ai) in enumerate(secs):   # zero taps are absent from the reference's sum anyway
      b[i, :len(bi)] = bi
      a[i, :len(ai)] = ai
    bank = FilterBank([(b, a)], n_inputs=1, mode="outer")
    # every filter received the same memory / zero (reference :1053) and applied the padding rule with its own
    # order: ``memories`` holds the staged list of every filter (None: all ``zero``)
    xh = np.full((len(secs), max(nb - 1, 1)), float(zero))
    yh = np.full((len(secs), max(na - 1, 1)), float(zero))
    for i, (_, ai) in enumerate(secs):
      lm = len(ai) - 1
      if memories[i] is not None:
        yh[i, :lm] = memory_to_hist(memories[i], lm, zero)
    bank.reset(zero=float(zero))
    bank.set_state(xh, yh)

    def gen():
      it = iter(seq)
      while True:
        chunk = list(itertools.islice(it, block))
        if not chunk:
          return
        x = np.asarray(chunk, dtype=np.float64).reshape(len(chunk), 1)
        y = bank.mixdown(bank.process(x, layout="time"), layout="time")
        yield y[:, 0].tolist()
    return Stream(itertools.chain.from_iterable(gen()))


# ---------------------------------------------------------------------------
# designs.  Formulas re-derived from the reference's docstrings / math/ notes and
# evaluated in the same operation order so that the doubles come out identical.
# ---------------------------------------------------------------------------
def _accepts_streams(design):
  """Give a scalar design its Stream-argument form (the reference's designs take Streams and
  return time-varying filters, :1179-1495): any iterable argument makes the coefficients Streams
  of the scalar design evaluated sample by sample (see timevar.design_over_streams)."""
  import functools

  @functools.wraps(design)
  def wrapper(*args, **kwargs):
    from . import timevar
    if kwargs:
      import inspect
      bound = inspect.signature(design).bind(*args, **kwargs)
      bound.apply_defaults()
      args = tuple(bound.arguments.values())
    if any(timevar.is_series(a) for a in args):
      return timevar.design_over_streams(design, *args)
    return design(*args)
  return wrapper


comb = StrategyDict("comb")


@comb.strategy("fb", "alpha", "fb_alpha", "feedback_alpha")
def comb(delay, alpha=1):
  """Feedback comb  y[n] = x[n] + alpha * y[n - delay]  (reference :1090-1118)."""
  return 1 / (1 - alpha * z ** -delay)


@comb.strategy("tau", "fb_tau", "feedback_tau")
def comb(delay, tau=float("inf")):
  """Feedback comb from the decay time tau, alpha = e ** (-delay / tau) (reference :1121-1147)."""
  alpha = math.e ** (-delay / tau)
  return 1 / (1 - alpha * z ** -delay)


@comb.strategy("ff", "ff_alpha", "feedforward_alpha")
def comb(delay, alpha=1):
  """Feedforward comb  y[n] = x[n] + alpha * x[n - delay]  (reference :1150-1173)."""
  return 1 + alpha * z ** -delay


# The designs below are written once for numbers and for Streams, like the reference's: the
# elementwise math helpers map over an iterable argument and ``thub`` lets a Stream-valued
# intermediate appear several times (its second argument = how many).  With Stream arguments the
# coefficients come out as Streams (a time-varying filter) built by cheap generator arithmetic --
# the same float operations, in the same order, as the scalar case.
def _elementwise(fn):
  def mapped(value):
    if hasattr(value, "__iter__"):
      from .stream import Stream
      return Stream(map(fn, iter(value)))
    return fn(value)
  return mapped


_exp, _cos, _sin, _sqrt = (_elementwise(f) for f in (math.exp, math.cos, math.sin, math.sqrt))


def _thub(value, uses):
  from .stream import thub
  return thub(value, uses)


resonator = StrategyDict("resonator")


@resonator.strategy("poles_exp")
def resonator(freq, bandwidth):
  """Two-pole resonator, 0 dB peak at the resonance (reference :1179-1209).
  ``freq`` and ``bandwidth`` in rad/sample; pole radius R = exp(-bandwidth / 2)."""
  R = _thub(_exp(-_thub(bandwidth, 1) * .5), 5)
  cost = _thub(_cos(freq) * (2 * R) / (1 + R ** 2), 2)
  gain = (1 - R ** 2) * _sqrt(1 - cost ** 2)
  return gain / (1 - 2 * R * cost * z ** -1 + R ** 2 * z ** -2)


@resonator.strategy("freq_poles_exp")
def resonator(freq, bandwidth):
  """Two-pole resonator with the poles exactly at ``freq`` (reference :1212-1242)."""
  R = _thub(_exp(-_thub(bandwidth, 1) * .5), 3)
  freq = _thub(freq, 2)
  gain = (1 - R ** 2) * _sin(freq)
  return gain / (1 - 2 * R * _cos(freq) * z ** -1 + R ** 2 * z ** -2)


@resonator.strategy("z_exp")
def resonator(freq, bandwidth):
  """Two poles plus zeros at DC and Nyquist, 0 dB at ``freq`` (reference :1245-1276)."""
  R = _thub(_exp(-_thub(bandwidth, 1) * .5), 5)
  cost = _cos(freq) * (1 + R ** 2) / (2 * R)
  gain = (1 - R ** 2) * .5
  return gain * (1 - z ** -2) / (1 - 2 * R * cost * z ** -1 + R ** 2 * z ** -2)


@resonator.strategy("freq_z_exp")
def resonator(freq, bandwidth):
  """As ``z_exp`` with the poles exactly at ``freq`` (reference :1279-1310)."""
  R = _thub(_exp(-_thub(bandwidth, 1) * .5), 3)
  gain = (1 - R ** 2) * .5
  return gain * (1 - z ** -2) / (1 - 2 * R * _cos(freq) * z ** -1 + R ** 2 * z ** -2)


lowpass = StrategyDict("lowpass")
highpass = StrategyDict("highpass")


def _one_pole_radius(x):
  x = _thub(x, 2)
  return _thub(x - _sqrt(x ** 2 - 1), 2)


@lowpass.strategy("pole")
def lowpass(cutoff):
  """One pole, exact -3 dB at ``cutoff`` rad/sample (reference :1370-1378)."""
  R = _one_pole_radius(2 - _cos(cutoff))
  return (1 - R) / (1 - R * z ** -1)


@highpass.strategy("pole")
def highpass(cutoff):
  """One pole highpass, mirror of lowpass.pole (reference :1381-1389)."""
  R = _one_pole_radius(2 + _cos(cutoff))
  return (1 - R) / (1 + R * z ** -1)


def _pole_zero_radius(num, cutoff):
  den = _cos(cutoff)
  if hasattr(den, "__iter__"):
    from .stream import Stream
    den = Stream((el if el else 1) for el in den)
  else:
    den = den if den else 1      # numerator already zero there (reference :1399-1403)
  return _thub(num / den, 2)


@lowpass.strategy("z")
def lowpass(cutoff):
  """One pole and a zero at Nyquist (reference :1392-1405)."""
  cutoff = _thub(cutoff, 2)
  R = _pole_zero_radius(_sin(cutoff) - 1, cutoff)
  gain = (1 + R) / 2
  return gain * (1 + z ** -1) / (1 + R * z ** -1)


@highpass.strategy("z")
def highpass(cutoff):
  """One pole and a zero at DC (reference :1408-1421)."""
  cutoff = _thub(cutoff, 2)
  R = _pole_zero_radius(1 - _sin(cutoff), cutoff)
  gain = (1 + R) / 2
  return gain * (1 - z ** -1) / (1 - R * z ** -1)


@lowpass.strategy("pole_exp")
def lowpass(cutoff):
  """Matched-Z one pole, R = e ** -cutoff (reference :1424-1437)."""
  R = _thub(_exp(-_thub(cutoff, 1)), 2)
  return (1 - R) / (1 - R * z ** -1)


@highpass.strategy("pole_exp")
def highpass(cutoff):
  """Matched-Z one pole highpass, R = e ** (cutoff - pi) (reference :1440-1454)."""
  R = _thub(_exp(_thub(cutoff, 1) - math.pi), 2)
  return (1 - R) / (1 + R * z ** -1)


@lowpass.strategy("z_exp")
def lowpass(cutoff):
  """Matched-Z pole plus zero at Nyquist (reference :1457-1472)."""
  R = _thub(_exp(_thub(cutoff, 1) - math.pi), 2)
  G = (R + 1) / 2
  return G * (1 + z ** -1) / (1 + R * z ** -1)


@highpass.strategy("z_exp")
def highpass(cutoff):
  """Matched-Z pole plus zero at DC (reference :1475-1490)."""
  R = _thub(_exp(-_thub(cutoff, 1)), 2)
  G = (R + 1) / 2
  return G * (1 - z ** -1) / (1 - R * z ** -1)


lowpass.default = lowpass.pole       # reference :1494
highpass.default = highpass.z        # reference :1495
comb.default = comb.fb
resonator.default = resonator.poles_exp
