"""The reference's CPython execution path, restated: TEST INFRASTRUCTURE ONLY.

Importable from tests/ and from the ``cpu_baseline`` leg of bench.py, never from audiolazy_amd/.

``LinearFilter.__call__`` (reference audiolazy/lazy_filters.py:141-264) does not interpret a
filter: it writes the difference equation as Python source, ``exec``s it (:98-106) and returns the
resulting generator wrapped in a Stream -- one CPython bytecode pass per sample.  This module
builds that source from coefficient lists with its own code (nothing is imported from the
reference) so that the GPU box, which has no /root/reference, can time *the same interpreter work*
beside the HIP engine:

  noise()             the cfg1 source: one ``random.uniform`` per sample  (lazy_synth.py:394-415)
  df1_source / df1    the generated generator                             (lazy_filters.py:197-260)
  consume_blocks      the ``.blocks(size)`` consumer                      (lazy_misc.py:74-129)

Parity status: pinned.  tests/test_oracle_golden.py holds ``df1`` to bit identity against the
reference-generated vectors of tests/golden/filters.json (the same vectors that pin
oracle/alz_oracle.c), for constant and for series coefficients.
"""
import collections
import itertools
import random


def _is_series(coef):
  return hasattr(coef, "__iter__")


def df1_source(b, a):
  """Source text of ``gen(seq, memory, zero, *series)`` for numerator ``b`` / denominator ``a``
  (lists indexed by delay; an entry is a number or an iterable giving one value per sample).

  Same statement per sample as the reference emits (lazy_filters.py:197-257): numerator terms by
  ascending delay, then denominator terms; a constant 0 contributes no term, +-1 no product; a
  series term is always present (``next(bK) * dK`` / ``-next(aK) * mK``); ``-( )`` for a gain of
  -1, ``( ) / gain`` for any other gain than 1; then the m and d variables shift, oldest first.
  Returns (source, names of the series arguments in call order)."""
  lb, la = len(b), len(a)
  terms, series = [], []
  for k, coef in enumerate(b):
    if _is_series(coef):
      series.append("b%d" % k)
      terms.append("next(b%d) * d%d" % (k, k))
    elif coef == 1:
      terms.append("d%d" % k)
    elif coef == -1:
      terms.append("-d%d" % k)
    elif coef != 0:
      terms.append("%s * d%d" % (format(coef), k))
  gain = 1
  for k, coef in enumerate(a):
    if _is_series(coef):
      series.append("a%d" % k)
      terms.append("-next(a%d) * m%d" % (k, k))
    elif k == 0:
      gain = coef
    elif coef == -1:
      terms.append("m%d" % k)
    elif coef == 1:
      terms.append("-m%d" % k)
    elif coef != 0:
      terms.append("-%s * m%d" % (format(coef), k))
  if not terms:
    return "def gen(seq, memory, zero):\n  for unused in seq:\n    yield zero\n", []
  total = " + ".join(terms)
  if gain == -1:
    total = "-(%s)" % total
  elif gain != 1:
    total = "(%s) / %s" % (total, format(gain))
  lines = ["def gen(%s):" % ", ".join(["seq", "memory", "zero"] + series)]
  if la > 1:
    lines.append("  %s = memory" % " ".join("m%d ," % k for k in range(1, la)))
  if lb > 1:
    lines.append("  %s = zero" % " = ".join("d%d" % k for k in range(1, lb)))
  lines += ["  for d0 in seq:", "    m0 = %s" % total, "    yield m0"]
  lines += ["    m%d = m%d" % (k, k - 1) for k in range(la - 1, 0, -1)]
  lines += ["    d%d = d%d" % (k, k - 1) for k in range(lb - 1, 0, -1)]
  return "\n".join(lines) + "\n", series


def df1(b, a, seq, memory=None, zero=0.):
  """The generator the reference would return for ``ZFilter(b, a)(seq, memory, zero)``.
  ``memory``: None or a sequence of past outputs, applied with the reference's rule (first
  ``len(a) - 1`` items, LEFT-padded with ``zero`` when short, :185-195)."""
  b = [c if _is_series(c) else (float(c) if isinstance(c, float) else c) for c in b]
  a = [c if _is_series(c) else (float(c) if isinstance(c, float) else c) for c in a]
  source, names = df1_source(b, a)
  scope = {}
  exec(source, scope)
  lm = len(a) - 1
  if memory is None:
    memory = [zero] * lm
  else:
    memory = list(itertools.islice(iter(memory), lm))
    memory = [zero] * (lm - len(memory)) + memory
  coefs = {"b%d" % k: c for k, c in enumerate(b)}
  coefs.update({"a%d" % k: c for k, c in enumerate(a)})
  return scope["gen"](iter(seq), memory, zero, *[iter(coefs[n]) for n in names])


def noise(n, seed):
  """``white_noise(n)``: n samples of ``random.uniform(-1., 1.)`` from a seeded generator."""
  rnd = random.Random(seed)
  uniform = rnd.uniform
  for _ in range(n):
    yield uniform(-1., 1.)


def consume_blocks(seq, size):
  """``for blk in Stream(seq).blocks(size)``: the reference's blockenizer with hop == size, one
  deque append and one comparison per sample, the same deque yielded per block.  Returns the
  number of samples that went through (tail block padded with 0., like the reference)."""
  window = collections.deque(maxlen=size)
  fill = seen = blocks = 0
  for item in seq:
    window.append(item)
    fill += 1
    if fill == size:
      blocks += 1           # (a consumer would look at ``window`` here)
      seen += size
      fill = 0
  if fill:
    seen += fill
    window.extend([0.] * (size - fill))
    blocks += 1
  return seen
