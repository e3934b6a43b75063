"""Time-varying linear filters: coefficients that are Streams, one value per output sample.

Host side of the Stream-coefficient branch of the reference's ``LinearFilter.__call__``
(audiolazy/lazy_filters.py:141-264): a coefficient that is an iterable contributes
``next(b_k) * d_k`` / ``-next(a_k) * m_k`` to the same left-to-right sum as the constant terms
(:197-224); a Stream ``a0`` is first normalised away by multiplying every other coefficient by
``1 / a0`` sample by sample (:166-174).  The coefficient streams are pulled ``block`` values at a
time -- exactly one value per input sample, like the reference's ``next`` calls -- uploaded, and
the recurrence runs on the GPU (``alz_tv_process_dev``); results are bit-identical.

``design_over_streams`` gives the reference's design functions their Stream-argument form
(:1179-1495 accept Streams): the coefficient streams are the scalar design evaluated sample by
sample, which is what the reference's elementwise Stream arithmetic computes.
"""
import ctypes
import itertools

import numpy as np

from . import _ffi


def is_series(v):
  """True for a coefficient that varies in time (any iterable, like the reference's check
  ``isinstance(coeff, Iterable)``, lazy_filters.py:202, :214)."""
  return hasattr(v, "__iter__")


def _pull(it, n):
  return list(itertools.islice(it, n))


class _Pool(object):
  """Device buffers kept from block to block (a hipMalloc per block and series would dominate
  small blocks): ``get(name, nbytes)`` returns a buffer of at least that size."""

  def __init__(self, device):
    self.device, self._bufs = device, {}

  def get(self, name, nbytes):
    buf = self._bufs.get(name)
    if buf is None or buf.nbytes < nbytes:
      buf = self._bufs[name] = _ffi.DevBuf(max(nbytes, 4096), self.device)
    return buf


def run(numlist, denlist, seq, memory=None, zero=0., block=None, device=0):
  """Iterator of output samples (see :func:`run_blocks`, whose per-block lists it chains: one
  Python-level resume per block instead of one per sample)."""
  return itertools.chain.from_iterable(run_blocks(numlist, denlist, seq, memory=memory, zero=zero, block=block,
                                                  device=device))


def run_blocks(numlist, denlist, seq, memory=None, zero=0., block=None, device=0):
  """Generator of lists of output samples, one list per block, for one input stream -- or, in the reference's vector-valued
  idiom, for C parallel streams: items of ``seq`` that are rows of C values, ``zero`` a row,
  ``memory`` a list of rows, and coefficient Streams whose items are numbers (shared by the
  channels) or rows (one coefficient per channel, e.g. ``repeat(ndarray)``).

  numlist / denlist : coefficients by delay (constants or iterables), denlist[0] = a0.

  Consumption: a block of ``block`` input items is pulled first, then one value per input item
  from every coefficient iterator.  When a coefficient iterator ends before the input does, the
  output ends there like the reference's, but up to one block of input has been consumed beyond
  that point (the reference's generator pulls exactly one extra input item, lazy_filters.py:251-253);
  use a small ``block_size`` when the leftover input matters.
  """
  from .bank import memory_to_hist, block_size
  L = _ffi.load()
  block = block_size() if block is None else block
  b = list(numlist) or [0.]
  a = list(denlist)
  if not a:
    raise ZeroDivisionError("Invalid filter gain")
  nb, na = len(b), len(a)
  b_it = [iter(v) if is_series(v) else None for v in b]
  a_it = [iter(v) if is_series(v) else None for v in a]
  if a_it[0] is None and a[0] == 0:
    raise ZeroDivisionError("Invalid filter gain")
  # With a constant gain the reference negates each denominator coefficient as it is read; done
  # here, in Python arithmetic, so that integer-valued coefficient streams give the reference's
  # signed zeros (ALZ_TV_NEGATED).  A series gain makes every coefficient a float product first.
  pre_negate = a_it[0] is None
  pool = _Pool(device)
  it = iter(seq)
  rows, C, d_xh, d_yh, zero_row = False, 1, None, None, None
  while True:
    chunk = _pull(it, block)
    if not chunk:
      return
    if d_xh is None:       # the first item tells scalars from rows
      rows = hasattr(chunk[0], "__len__")
      C = len(chunk[0]) if rows else 1
      zero_row = np.broadcast_to(np.asarray(zero, dtype=np.float64), (C,)).copy()
      hist = memory_to_hist(memory, na - 1, zero)
      xh = np.tile(zero_row, (max(nb - 1, 1), 1))
      yh = np.zeros((max(na - 1, 1), C))
      for k, item in enumerate(hist):
        yh[k] = np.broadcast_to(np.asarray(item, dtype=np.float64), (C,))
      d_xh = _ffi.DevBuf(xh.nbytes, device).upload(xh)
      d_yh = _ffi.DevBuf(yh.nbytes, device).upload(yh)
    n = len(chunk)
    series = {}
    for side, its in (("b", b_it), ("a", a_it)):
      for k, src in enumerate(its):
        if src is not None:
          vals = _pull(src, n)
          n = min(n, len(vals))      # a coefficient stream that ends, ends the output
          if side == "a" and k > 0 and pre_negate:
            vals = [-v for v in vals]   # ``-next(a_k)`` in Python arithmetic: an int 0 stays 0, not -0.0
          series[(side, k)] = vals
    if n == 0:
      return
    x = np.ascontiguousarray(np.asarray(chunk[:n], dtype=np.float64).reshape(n, C))
    cols = {key: np.asarray(vals[:n], dtype=np.float64).reshape(n, -1) for key, vals in series.items()}
    gain = a[0]
    if ("a", 0) in cols:             # series a0: every other coefficient times 1 / a0 (:166-174)
      inv = 1.0 / cols.pop(("a", 0))
      for k in range(nb):
        if ("b", k) in cols:
          cols[("b", k)] = cols[("b", k)] * inv
        elif np.any(np.asarray(b[k]) != 0):
          cols[("b", k)] = np.asarray(b[k], dtype=np.float64) * inv
      for k in range(1, na):
        if ("a", k) in cols:
          cols[("a", k)] = cols[("a", k)] * inv
        elif np.any(np.asarray(a[k]) != 0):
          cols[("a", k)] = np.asarray(a[k], dtype=np.float64) * inv
      gain = 1.0
    bufs = {}
    for key, arr in cols.items():
      if arr.shape[1] not in (1, C):
        raise ValueError("a coefficient row has %d values for %d channels" % (arr.shape[1], C))
      arr = np.ascontiguousarray(arr)
      bufs[key] = (pool.get(key, arr.nbytes).upload(arr), arr.shape[1])

    def taps(side, count, consts):
      arr = (_ffi.TvTap * count)()
      for k in range(count):
        if (side, k) in bufs:
          buf, width = bufs[(side, k)]
          negated = _ffi.TV_NEGATED if (side == "a" and k > 0 and pre_negate) else 0
          arr[k] = _ffi.TvTap(0.0, buf.ptr.value, width, 1 if (width == C and C > 1) else 0, negated)
        else:
          arr[k] = _ffi.TvTap(float(consts[k]), None, 0, 0, 0)
      return arr
    tb = taps("b", nb, b)
    ta = taps("a", na, [gain] + list(a[1:]))
    d_x = pool.get("x", x.nbytes).upload(x)
    d_y = pool.get("y", x.nbytes)
    _ffi.check(L.alz_tv_process_dev(nb, ctypes.cast(tb, ctypes.c_void_p), na, ctypes.cast(ta, ctypes.c_void_p), C,
                                    d_x.ptr, d_y.ptr, n, _ffi.TIME_MAJOR, C, C, d_xh.ptr, d_yh.ptr,
                                    float(zero_row[0]), device, None))
    _ffi.check(L.alz_device_sync(device))
    y = d_y.download((n, C), np.float64)
    yield list(y) if rows else y[:, 0].tolist()      # one block of results (the caller chains the blocks)
    if n < len(chunk):
      return


def process_block(b, a, x, xh=None, yh=None, zero=0., layout="time"):
  """Array-level entry: one block of C channels through a time-varying filter.

  b / a : per delay, a float or a float64 torch CUDA tensor of shape [N] (shared by all channels)
          or matching ``x`` (per channel).  x : float64 CUDA tensor [N, C] ("time") or [C, N].
  xh / yh : [nb-1, C] / [na-1, C] CUDA tensors, updated in place (None: zeros, not returned).
  """
  import torch
  L = _ffi.load()
  lay = _ffi.TIME_MAJOR if layout == "time" else _ffi.CHAN_MAJOR
  if not x.is_cuda or x.dtype != torch.float64 or not x.is_contiguous() or x.dim() != 2:
    raise ValueError("x must be a contiguous 2-D float64 CUDA tensor")
  n, C = (tuple(x.shape) if lay == _ffi.TIME_MAJOR else tuple(x.shape)[::-1])
  nb, na = len(b), len(a)
  keep = []

  def taps(coefs):
    arr = (_ffi.TvTap * len(coefs))()
    for k, v in enumerate(coefs):
      if type(v).__module__.startswith("torch"):
        if not v.is_cuda or v.dtype != torch.float64 or not v.is_contiguous():
          raise ValueError("coefficient series must be contiguous float64 CUDA tensors")
        keep.append(v)
        if v.dim() == 1:
          if v.numel() != n:
            raise ValueError("a shared coefficient series needs one value per sample")
          arr[k] = _ffi.TvTap(0.0, v.data_ptr(), 1, 0, 0)
        elif tuple(v.shape) == tuple(x.shape):
          arr[k] = _ffi.TvTap(0.0, v.data_ptr(), *((C, 1) if lay == _ffi.TIME_MAJOR else (1, n)), 0)
        else:
          raise ValueError("a per-channel coefficient series must have the shape of x")
      else:
        arr[k] = _ffi.TvTap(float(v), None, 0, 0, 0)
    return arr
  tb, ta = taps(b), taps(a)
  if xh is None:
    xh = torch.full((max(nb - 1, 1), C), float(zero), dtype=torch.float64, device=x.device)
  if yh is None:
    yh = torch.full((max(na - 1, 1), C), float(zero), dtype=torch.float64, device=x.device)
  y = torch.empty_like(x)
  ld = C if lay == _ffi.TIME_MAJOR else n
  stream = torch.cuda.current_stream(x.device).cuda_stream
  _ffi.check(L.alz_tv_process_dev(nb, ctypes.cast(tb, ctypes.c_void_p), na, ctypes.cast(ta, ctypes.c_void_p), C,
                                  x.data_ptr(), y.data_ptr(), n, lay, ld, ld, xh.data_ptr(), yh.data_ptr(),
                                  float(zero), x.device.index or 0, ctypes.c_void_p(stream)))
  return y


def design_over_streams(design, *args):
  """The Stream-argument form of a scalar design function: ``design(*scalars)`` must return a
  filter with constant coefficients (a ZFilter, or a CascadeFilter of them); the result has the
  same structure with coefficients that are Streams holding that design evaluated for every
  sample of the argument streams (scalars among the arguments are held constant; the stream ends
  with the shortest argument)."""
  from .filters import ZFilter, CascadeFilter
  from .stream import Stream
  iters = [iter(a) if is_series(a) else itertools.repeat(a) for a in args]
  first_args = [next(it) for it in iters]
  first_filt = design(*first_args)
  cascade = isinstance(first_filt, CascadeFilter)
  first_secs = list(first_filt) if cascade else [first_filt]
  # Which coefficients exist is a property of the design, not of its first sample (lowpass.z at
  # exactly pi/2 has R == 0): the structure is the union of the first sample's and that of a
  # generic probe point next to it (only the streamed arguments are moved).
  probe_secs = first_secs
  try:
    probe = design(*[v * 0.9371 + 0.0113 if is_series(a) else v for a, v in zip(args, first_args)])
    probe_secs = list(probe) if cascade else [probe]
  except (ValueError, ZeroDivisionError, ArithmeticError):
    pass

  def coef_at(filt, side, k):
    lst = filt.numlist if side == "b" else filt.denlist
    return lst[k] if k < len(lst) else 0.0
  shapes = [(max(len(f.numlist), len(g.numlist)), max(len(f.denlist), len(g.denlist)))
            for f, g in zip(first_secs, probe_secs)]
  n_coefs = sum(nb + na for nb, na in shapes)

  def designs():
    yield first_secs
    for vals in zip(*iters):
      filt = design(*vals)
      yield list(filt) if cascade else [filt]
  shared = itertools.tee(designs(), max(n_coefs, 1))

  def coef(src, sec, side, k):
    for secs in src:
      yield coef_at(secs[sec], side, k)
  out, idx = [], 0
  for sec, (nb, na) in enumerate(shapes):
    num, den = [], []
    for k in range(nb):
      # structural constants stay constants (absent taps, a0 = 1), as in the reference's algebra
      absent = coef_at(first_secs[sec], "b", k) == 0 and coef_at(probe_secs[sec], "b", k) == 0
      num.append(0.0 if absent else Stream(coef(shared[idx], sec, "b", k)))
      idx += 1
    for k in range(na):
      v, w = coef_at(first_secs[sec], "a", k), coef_at(probe_secs[sec], "a", k)
      const = (v == 0 and w == 0) or (k == 0 and v == 1 and w == 1)
      den.append(v if const else Stream(coef(shared[idx], sec, "a", k)))
      idx += 1
    out.append(ZFilter(num, den))
  return CascadeFilter(out) if cascade else out[0]
