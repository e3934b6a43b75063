"""FilterBank: the host side of the blocked stream-filter engine.

Mirrors the reference's callable protocol -- "a filter is any callable that
receives an iterable as input and returns a Stream" with the signature
``__call__(seq, memory=None, zero=0.)`` (reference audiolazy/lazy_filters.py
:141, :840, :975-978) -- for a whole bank of independent channels, and adds an
array-in / array-out ``process`` for blocks already held as arrays.

Design stays on the host in float64 (coefficients come from any object with
``numlist`` / ``denlist``, reference lazy_filters.py:55-67, or from raw arrays);
execution is always libalzhip.so (HIP, gfx950).  No CPU execution path exists
here: without the library or a GPU the calls raise.
"""
import ctypes
import itertools
import weakref

import numpy as np

from . import _ffi

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)

# Items pulled from the input (and from coefficient Streams) per launch by the filter call
# protocol.  Large blocks amortise the launch; interactive uses (a ControlStream steering a
# time-varying filter, audio played as it is made) want a small one: ``block_size(64)``.
_block = [4096]


def block_size(n=None):
  """Get / set the number of items every ``filt(seq)`` call pulls per GPU launch."""
  if n is not None:
    if int(n) < 1:
      raise ValueError("block size must be positive")
    _block[0] = int(n)
  return _block[0]


LAYOUTS = {"time": _ffi.TIME_MAJOR, "chan": _ffi.CHAN_MAJOR,
           _ffi.TIME_MAJOR: _ffi.TIME_MAJOR, _ffi.CHAN_MAJOR: _ffi.CHAN_MAJOR}


def _is_torch(x):
  return type(x).__module__.startswith("torch")


def _coef_lists(filt):
  """(numlist, denlist) of one section as float lists.

  Raises what the reference raises: ValueError("Non-causal filter") comes out
  of the filter's own numlist/denlist properties (lazy_filters.py:55-67);
  Stream-valued (time-varying) coefficients are outside the engine's gate.
  """
  if isinstance(filt, (tuple, list)) and len(filt) == 2 and not hasattr(filt, "numlist"):
    b, a = filt
  else:
    b, a = filt.numlist, filt.denlist
  b, a = list(b), list(a)
  for v in itertools.chain(b, a):
    if hasattr(v, "__iter__") and not isinstance(v, np.ndarray):
      raise NotImplementedError("time-varying (Stream) coefficients are outside the engine's gate")
  return b, a


def sections_of(filt):
  """Flatten a filter object into its cascade sections [(b, a), ...].

  A CascadeFilter-like object (a list of filters, reference lazy_filters.py
  :970-1021) contributes one section per item, in call order (:988-990).
  """
  # (a CascadeFilter IS a list and, like every linear filter of the reference, also has numlist / denlist -- of the
  #  PRODUCT polynomial: its sections are its members, never that product)
  if not isinstance(filt, list) and (hasattr(filt, "numlist") or (isinstance(filt, tuple) and len(filt) == 2)):
    return [_coef_lists(filt)]
  if isinstance(filt, (list, tuple)):
    out = []
    for f in filt:
      out.extend(sections_of(f))
    return out
  raise TypeError("cannot read filter coefficients from %r" % (type(filt),))


def memory_to_hist(memory, lm, zero=0.0):
  """memory -> [m1 .. m_lm], the reference's rule (lazy_filters.py:185-195).

  None -> [zero]*lm; callable -> memory(lm); iterable -> its first lm items,
  LEFT-padded with ``zero`` when short (zero_pad's second positional argument
  is ``left``, lazy_misc.py:132 -- a quirk reproduced on purpose).
  """
  if memory is None:
    return [zero] * lm
  if not hasattr(memory, "__iter__"):
    memory = memory(lm)
  out = list(itertools.islice(iter(memory), lm)) if lm > 0 else []
  if len(out) < lm:
    out = [zero] * (lm - len(out)) + out
  return out


def stage_memories(memory, nas, zero=0.0):
  """``memory`` as every cascade stage sees it: one [m1 .. m_lm] list per stage (each stage gets
  the same argument, reference lazy_filters.py:989, and applies the padding rule with its own
  order; a callable is called once per stage).  None stays None.  Evaluated at call time, like
  the reference evaluates ``memory`` before it builds its generator (:185-195)."""
  if memory is None:
    return None
  if callable(memory) and not hasattr(memory, "__iter__"):
    return [memory_to_hist(memory, na - 1, zero) for na in nas]
  items = list(itertools.islice(iter(memory), max(max(nas) - 1, 0)))
  return [memory_to_hist(items, na - 1, zero) for na in nas]


def call_sections(sections, seq, memory=None, zero=0., block=None, input_map=None, _hists=None, output_map=None):
  """The filter call protocol for one cascade of LTI sections [(b, a), ...]: ``memory`` is read
  now, the input is not touched until the result is iterated (the reference's generator pulls
  its first sample at the first ``next``); the first item then tells whether samples are scalars
  or rows of C values (C parallel streams, reference tests/test_filters_extdep.py:49-89).
  ``input_map`` / ``output_map``: elementwise stages of :mod:`audiolazy_amd.maps` on the device around the
  filter, per BLOCK ("abs" / "square" in front; "sqrt" behind -- ``envelope``'s root)."""
  from .stream import Stream
  hists = stage_memories(memory, [len(a) for _, a in sections], zero) if _hists is None else \
      [memory_to_hist(h, len(a) - 1, zero) for h, (_, a) in zip(_hists, sections)]

  def blocks_out():
    it = iter(seq)
    for first in it:
      break
    else:
      return
    n_inputs = len(first) if hasattr(first, "__len__") else 1
    bank = FilterBank(sections, n_inputs=n_inputs)
    if input_map:
      bank.set_input_map(input_map)
    bank.reset(zero=zero, _hists=hists)
    for out in bank._blocks(itertools.chain([first], it), block, post=output_map):
      yield out
  # one Python-level resume per BLOCK, not per sample: the items of a block come out of a list
  return Stream(itertools.chain.from_iterable(blocks_out()))


def mix_sets(y, n_sets, n_inputs, layout="time", out=None, device=0):
  """Ordered sum over the leading ``n_sets`` groups of ``n_inputs`` channels (alz_mix_dev)."""
  L = _ffi.load()
  lay = LAYOUTS[layout]
  if _is_torch(y):
    import torch
    if not y.is_cuda or y.dtype != torch.float64 or not y.is_contiguous() or y.dim() != 2:
      raise ValueError("torch input must be a contiguous 2-D float64 CUDA tensor")
    n, ch = (tuple(y.shape) if lay == _ffi.TIME_MAJOR else tuple(y.shape)[::-1])
    if ch != n_sets * n_inputs:
      raise ValueError("block has %d channels, expected %d sets x %d inputs" % (ch, n_sets, n_inputs))
    shape = (n, n_inputs) if lay == _ffi.TIME_MAJOR else (n_inputs, n)
    res = torch.empty(shape, dtype=torch.float64, device=y.device) if out is None else out
    ldy, ldo = (ch, n_inputs) if lay == _ffi.TIME_MAJOR else (n, n)
    stream = torch.cuda.current_stream(y.device).cuda_stream
    _ffi.check(L.alz_mix_dev(y.data_ptr(), n_sets, n_inputs, n, lay, ldy, ldo, res.data_ptr(),
                             y.device.index or 0, ctypes.c_void_p(stream)))
    return res
  y = np.ascontiguousarray(y, dtype=np.float64)
  n, ch = (y.shape if lay == _ffi.TIME_MAJOR else y.shape[::-1])
  if ch != n_sets * n_inputs:
    raise ValueError("block has %d channels, expected %d sets x %d inputs" % (ch, n_sets, n_inputs))
  shape = (n, n_inputs) if lay == _ffi.TIME_MAJOR else (n_inputs, n)
  if n == 0:
    return np.empty(shape)
  ldy, ldo = (ch, n_inputs) if lay == _ffi.TIME_MAJOR else (n, n)
  d_y = _ffi.DevBuf(y.nbytes, device).upload(y)
  d_o = _ffi.DevBuf(n * n_inputs * 8, device)
  _ffi.check(L.alz_mix_dev(d_y.ptr, n_sets, n_inputs, n, lay, ldy, ldo, d_o.ptr, device, None))
  _ffi.check(L.alz_device_sync(device))
  res = d_o.download(shape, np.float64)
  if out is not None:
    out[...] = res
    return out
  return res


def mix_tracks(tracks, deltas, zero=0., device=0):
  """Streamix for tracks that are arrays, summed on the GPU (alz_mix_tracks_dev): ``tracks[k]``
  enters ``deltas[k]`` samples after ``tracks[k-1]`` (Streamix.add's clock, floats allowed) and
  the tracks playing at a sample are added in the order given, starting from ``zero``.  The
  result is what ``list(smix)`` gives for ``smix.add(deltas[k], tracks[k])``, bit for bit.

  tracks : 1-D float64 NumPy arrays (returns a NumPy array) or contiguous float64 torch CUDA
           tensors (returns a CUDA tensor)."""
  from .stream import _mix_starts
  L = _ffi.load()
  starts = _mix_starts(deltas)
  if len(starts) != len(tracks):
    raise ValueError("one delta per track")
  use_torch = len(tracks) > 0 and _is_torch(tracks[0])
  lengths = [int(t.numel() if use_torch else np.asarray(t).size) for t in tracks]
  # (an empty track adds nothing to the sum, but the mixer keeps yielding ``zero`` until it starts)
  n_out = max([s + n for s, n in zip(starts, lengths)] + [0])
  order = [k for k in range(len(tracks)) if lengths[k] > 0]
  nt = len(order)
  c_starts = (ctypes.c_int64 * max(nt, 1))(*[starts[k] for k in order])
  c_lengths = (ctypes.c_int64 * max(nt, 1))(*[lengths[k] for k in order])
  ptrs = (ctypes.c_void_p * max(nt, 1))()
  if use_torch:
    import torch
    dev = tracks[0].device
    for i, k in enumerate(order):
      t = tracks[k]
      if not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous():
        raise ValueError("tracks must be contiguous float64 CUDA tensors")
      ptrs[i] = t.data_ptr()
    out = torch.empty((n_out,), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _ffi.check(L.alz_mix_tracks_dev(nt, ptrs, c_starts, c_lengths, float(zero), n_out, out.data_ptr(),
                                    dev.index or 0, ctypes.c_void_p(stream)))
    return out
  bufs = []
  for i, k in enumerate(order):
    arr = np.ascontiguousarray(tracks[k], dtype=np.float64).reshape(-1)
    bufs.append(_ffi.DevBuf(arr.nbytes, device).upload(arr))
    ptrs[i] = bufs[-1].ptr.value
  if n_out == 0:
    return np.empty((0,))
  d_out = _ffi.DevBuf(n_out * 8, device)
  _ffi.check(L.alz_mix_tracks_dev(nt, ptrs, c_starts, c_lengths, float(zero), n_out, d_out.ptr, device, None))
  _ffi.check(L.alz_device_sync(device))
  return d_out.download((n_out,), np.float64)


class FilterBank(object):
  """``channels`` independent streams through per-channel cascades.

  Parameters
  ----------
  sections :
    list of (b, a) per cascade section; ``b`` is [nb] (shared) or
    [n_sets, nb], ``a`` likewise -- the reference's numlist / denlist.
  n_inputs :
    number of input channels.  ``mode="diagonal"``: channel c filters input c
    with coefficient set c (or the single shared set).  ``mode="outer"``:
    every coefficient set runs on every input, channel = set * n_inputs +
    input (a filterbank such as the gammatone bank).
  """

  def __init__(self, sections, n_inputs=None, mode="diagonal", device=0):
    L = _ffi.load()
    secs = []
    n_sets = 1
    for b, a in sections:
      b = np.atleast_1d(np.asarray(b, dtype=np.float64))
      a = np.atleast_1d(np.asarray(a, dtype=np.float64))
      if b.shape[-1] == 0:
        b = np.zeros(b.shape[:-1] + (1,))
      if a.shape[-1] == 0:
        raise ZeroDivisionError("Invalid filter gain")  # empty denominator == 0
      for arr in (b, a):
        if arr.ndim == 2 and arr.shape[0] != 1:
          if n_sets not in (1, arr.shape[0]):
            raise ValueError("sections disagree on the number of coefficient sets")
          n_sets = arr.shape[0]
      secs.append((b, a))
    if mode not in ("diagonal", "outer"):
      raise ValueError("mode must be 'diagonal' or 'outer'")
    if n_inputs is None:
      n_inputs = n_sets if mode == "diagonal" else 1
    if mode == "diagonal" and n_sets not in (1, n_inputs):
      raise ValueError("diagonal bank: %d coefficient sets for %d inputs" % (n_sets, n_inputs))
    self.n_sets, self.n_inputs, self.mode, self.device = int(n_sets), int(n_inputs), mode, int(device)
    self.channels = self.n_inputs if mode == "diagonal" else self.n_sets * self.n_inputs
    self.nb = [s[0].shape[-1] for s in secs]
    self.na = [s[1].shape[-1] for s in secs]
    self.thx = sum(n - 1 for n in self.nb)
    self.thy = sum(n - 1 for n in self.na)

    def rows(arr):
      arr = arr.reshape(1, -1) if arr.ndim == 1 else arr
      return np.broadcast_to(arr, (self.n_sets, arr.shape[1]))

    self.b = np.ascontiguousarray(np.concatenate([rows(s[0]) for s in secs], axis=1))
    self.a = np.ascontiguousarray(np.concatenate([rows(s[1]) for s in secs], axis=1))
    nb = (ctypes.c_int * len(secs))(*self.nb)
    na = (ctypes.c_int * len(secs))(*self.na)
    handle = ctypes.c_void_p()
    _ffi.check(L.alz_bank_create(self.n_sets, self.n_inputs,
                                 _ffi.BANK_DIAGONAL if mode == "diagonal" else _ffi.BANK_OUTER,
                                 len(secs), nb, na, self.b.ctypes.data_as(_dp),
                                 self.a.ctypes.data_as(_dp), self.device, ctypes.byref(handle)))
    self._h = handle
    self._L = L
    self._live = None      # weak reference to the generator of the latest __call__
    self._fused = False
    self._time_parallel = 0
    self._input_map = None

  # -- construction helpers ------------------------------------------------
  @classmethod
  def from_filters(cls, filters, **kwargs):
    """One channel per filter object (ZFilter / CascadeFilter duck-typed):
    all must share the section shapes; coefficients become per-channel sets."""
    per = [sections_of(f) for f in filters]
    shapes = [[(len(b), len(a)) for b, a in p] for p in per]
    n_sec = len(per[0])
    if any(len(p) != n_sec for p in per):
      raise ValueError("filters disagree on the number of cascade sections")
    secs = []
    for s in range(n_sec):
      nb = max(sh[s][0] for sh in shapes)
      na = max(sh[s][1] for sh in shapes)
      # trailing zero taps are absent from the reference's expression anyway
      b = np.zeros((len(per), max(nb, 1)))
      a = np.zeros((len(per), na))
      for i, p in enumerate(per):
        b[i, :len(p[s][0])] = p[s][0]
        a[i, :len(p[s][1])] = p[s][1]
      secs.append((b, a))
    return cls(secs, **kwargs)

  def __del__(self):
    h, self._h = getattr(self, "_h", None), None
    if h:
      try:
        self._L.alz_bank_destroy(h)
      except Exception:
        pass

  # -- state ---------------------------------------------------------------
  def reset(self, memory=None, zero=0.0, _hists=None):
    """Start a new stream: the reference's ``memory`` / ``zero`` call arguments
    (lazy_filters.py:149-157, 185-195, 243-250), forwarded unchanged to every
    cascade stage like CascadeFilter does (:989)."""
    zero_arr = np.asarray(zero, dtype=np.float64)
    if _hists is None:
      _hists = stage_memories(memory, self.na, zero)
    if _hists is None and zero_arr.ndim == 0:
      _ffi.check(self._L.alz_bank_reset(self._h, float(zero)))
      return
    _ffi.check(self._L.alz_bank_reset(self._h, float(zero_arr.ravel()[0]) if zero_arr.size else 0.0))
    C = self.channels
    zc = np.broadcast_to(zero_arr, (C,)) if zero_arr.ndim else np.full(C, float(zero_arr))
    xh = np.repeat(zc[:, None], max(self.thx, 1), axis=1).copy()
    yh = np.empty((C, max(self.thy, 1)))
    off = 0
    for s_i, na in enumerate(self.na):
      hist = [zero] * (na - 1) if _hists is None else _hists[s_i]
      for k, item in enumerate(hist):
        yh[:, off + k] = np.broadcast_to(np.asarray(item, dtype=np.float64), (C,))
      off += na - 1
    self.set_state(xh, yh)

  def set_state(self, xh, yh):
    xh = np.ascontiguousarray(xh, dtype=np.float64)
    yh = np.ascontiguousarray(yh, dtype=np.float64)
    if xh.shape != (self.channels, max(self.thx, 1)) and xh.shape != (self.channels, self.thx):
      raise ValueError("xh must be [channels, sum(nb-1)]")
    if yh.shape != (self.channels, max(self.thy, 1)) and yh.shape != (self.channels, self.thy):
      raise ValueError("yh must be [channels, sum(na-1)]")
    _ffi.check(self._L.alz_bank_set_state(self._h, xh.ctypes.data_as(_dp), yh.ctypes.data_as(_dp)))

  def get_state(self):
    xh = np.zeros((self.channels, max(self.thx, 1)))
    yh = np.zeros((self.channels, max(self.thy, 1)))
    _ffi.check(self._L.alz_bank_get_state(self._h, xh.ctypes.data_as(_dp), yh.ctypes.data_as(_dp)))
    return xh[:, :self.thx], yh[:, :self.thy]

  def set_fused(self, on=True):
    """Opt into FMA contraction in the streaming biquad kernel: faster (the recurrence chain is
    two fused ops instead of four), no longer bit-identical to the reference (differences around
    1e-13 normalised; the contract is 1e-6).  Off by default."""
    _ffi.check(self._L.alz_bank_set_fused(self._h, 1 if on else 0))
    self._fused = bool(on)
    return self

  def set_input_map(self, op=None):
    """Apply an elementwise stage to every input sample before the first section sees it:
    ``"abs"`` (``filt(abs(sig))``, the shape of envelope.abs, reference lazy_analysis.py:468-493),
    ``"neg"``, ``"square"`` (``x * x``: not the reference's ``x ** 2`` in the last bit, see
    :mod:`audiolazy_amd.maps`) or None.  ``"abs"`` in front of one biquad-class section is fused
    into the kernels' input reads; everything else costs one extra streaming pass."""
    code = 0 if op is None else _ffi.MAP_OPS[op]
    _ffi.check(self._L.alz_bank_set_input_map(self._h, code))
    self._input_map = op
    return self

  def set_time_parallel(self, chunk=True):
    """Opt into the time-parallel mode for narrow banks (alz_bank_set_time_parallel): the time axis
    of every block is cut into chunks that run side by side (zero-state pass, propagation of the
    chunk states, replay).  ``chunk``: True / -1 = form and chunk length chosen by the engine (from
    256 channels up on time-major blocks the ONE-PASS form: 512-sample chunks resident in LDS, the
    block read once), False / 0 = off, a positive int = samples per chunk (three launches),
    "one-pass" / -2 = the one-pass form wherever it applies (single biquad-class sections).
    Not bit-identical to the reference (the contract's 1e-6 with orders of magnitude to spare); off
    by default."""
    n = -2 if chunk == "one-pass" else -1 if chunk is True else 0 if not chunk else int(chunk)
    _ffi.check(self._L.alz_bank_set_time_parallel(self._h, n))
    self._time_parallel = n
    return self

  def set_look_check(self, mode="call"):
    """What happens when the one-pass form of the time-parallel mode gives up one of its bounded waits (its workgroups
    wait for each other; foreign work on the device can keep some of them from starting) -- alz_bank_set_look_check.
    ``"call"`` (default): :meth:`process` waits for its own work and, if the kernel gave up, puts the bank's state back
    and processes the block again with the three-launch form (an in-place block raises instead: its input is gone).
    ``"deferred"``: :meth:`process` stays asynchronous and the NEXT call on the bank raises."""
    code = {"call": 1, "deferred": 0}[mode]
    _ffi.check(self._L.alz_bank_set_look_check(self._h, code))
    self._look_check = mode
    return self

  @property
  def look_stats(self):
    """Counters of the one-pass time-parallel kernel on this bank: launches, launches that gave up a wait, blocks
    processed again because of that, names of the waits that ran out last (alz_bank_look_stats)."""
    a, b, c, m = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_uint()
    _ffi.check(self._L.alz_bank_look_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(m)))
    return {"launches": a.value, "gave_up": b.value, "reruns": c.value,
            "last_sites": [n for k, n in enumerate(_ffi.LOOK_WAIT_SITES) if m.value >> k & 1]}

  @property
  def last_kernel(self):
    return self._L.alz_bank_last_kernel(self._h).decode()

  # -- execution -----------------------------------------------------------
  def process(self, x, layout="time", out=None):
    """Filter one block, continuing from the current state.

    x : [N, n_inputs] (layout="time") or [n_inputs, N] (layout="chan"), float64;
        a NumPy array (staged through the device) or a torch CUDA tensor
        (processed in place in HBM on the current torch stream).
    Returns an array / tensor of [N, channels] or [channels, N].
    """
    lay = LAYOUTS[layout]
    if _is_torch(x):
      return self._process_torch(x, lay, out)
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.ndim == 1:
      x = x.reshape(-1, 1) if lay == _ffi.TIME_MAJOR else x.reshape(1, -1)
    n, cin = (x.shape if lay == _ffi.TIME_MAJOR else x.shape[::-1])
    if cin != self.n_inputs:
      raise ValueError("block has %d input channels, bank expects %d" % (cin, self.n_inputs))
    shape = (n, self.channels) if lay == _ffi.TIME_MAJOR else (self.channels, n)
    y = np.empty(shape) if out is None else out
    if n == 0:
      return y
    ldx = self.n_inputs if lay == _ffi.TIME_MAJOR else n
    ldy = self.channels if lay == _ffi.TIME_MAJOR else n
    _ffi.check(self._L.alz_bank_process_host(self._h, x.ctypes.data_as(_dp), y.ctypes.data_as(_dp),
                                             n, lay, ldx, ldy))
    return y

  @staticmethod
  def _row_pitch(t, what):
    """Leading dimension (elements between consecutive rows) of a 2-D float64 CUDA tensor whose rows are contiguous:
    a contiguous tensor, or a view of a wider one (``base[:, :cols]``: rows padded -- the C ABI's ldx / ldy).  A row
    pitch that is a large power of two (rows of 2^18 .. 2^20 doubles) maps consecutive rows onto the same HBM channel:
    channel-major blocks with such rows gain 3 - 16 % from a pad of 32 doubles (profiles/r05_pitch_probe.log)."""
    import torch
    if not t.is_cuda or t.dtype != torch.float64 or t.dim() != 2:
      raise ValueError("torch %s must be a 2-D float64 CUDA tensor" % what)
    rows, cols = t.shape
    if t.is_contiguous():
      return cols
    if (cols > 1 and t.stride(1) != 1) or (rows > 1 and t.stride(0) < cols):
      raise ValueError("torch %s must have contiguous rows (a contiguous tensor or a column-sliced view of one)" % what)
    return t.stride(0) if rows > 1 else cols

  def _process_torch(self, x, lay, out):
    import torch
    ldx = self._row_pitch(x, "input")
    n, cin = (tuple(x.shape) if lay == _ffi.TIME_MAJOR else tuple(x.shape)[::-1])
    if cin != self.n_inputs:
      raise ValueError("block has %d input channels, bank expects %d" % (cin, self.n_inputs))
    shape = (n, self.channels) if lay == _ffi.TIME_MAJOR else (self.channels, n)
    y = torch.empty(shape, dtype=torch.float64, device=x.device) if out is None else out
    if n == 0:
      return y
    if tuple(y.shape) != shape or y.device != x.device:
      raise ValueError("out must be a %s float64 CUDA tensor on the input's device" % (shape,))
    ldy = self._row_pitch(y, "out")
    stream = torch.cuda.current_stream(x.device).cuda_stream
    _ffi.check(self._L.alz_bank_process_dev(self._h, x.data_ptr(), y.data_ptr(), n, lay, ldx, ldy,
                                            ctypes.c_void_p(stream)))
    return y

  def sync(self):
    _ffi.check(self._L.alz_bank_sync(self._h))

  def mixdown(self, y, layout="time", out=None):
    """Sum an OUTER bank's output block over its coefficient sets, ``((y_0 + y_1) + y_2) ...`` in
    set order -- ParallelFilter's sum (reference lazy_filters.py:1048-1054), on the device.

    y : [N, n_sets * n_inputs] / [n_sets * n_inputs, N] as returned by :meth:`process`.
    Returns [N, n_inputs] / [n_inputs, N] (same array kind as ``y``)."""
    if self.mode != "outer":
      raise ValueError("mixdown sums over the coefficient sets of an OUTER bank")
    return mix_sets(y, self.n_sets, self.n_inputs, layout=layout, out=out, device=self.device)

  def __call__(self, seq, memory=None, zero=0., block=None):
    """The reference's filter call: any iterable in, a Stream out.

    Items of ``seq`` are scalars (one input channel) or rows of ``n_inputs``
    values (the reference's vector-valued-sample idiom); output items are
    scalars when the bank has one channel, rows of ``channels`` otherwise.
    The input is pulled ``block`` items at a time (lazy per block, not per
    sample).
    """
    from .stream import Stream
    # every call owns its state, like every reference call owns its generator's locals: while a
    # Stream of an earlier call is still alive the new call runs on a copy of the bank
    live = self._live() if self._live is not None else None
    runner = self._clone() if (live is not None and live.gi_frame is not None) else self
    runner.reset(memory=memory, zero=zero)
    g = runner._blocks(seq, block)
    runner._live = weakref.ref(g)
    return Stream(itertools.chain.from_iterable(g))

  def _clone(self):
    """A bank with the same coefficients and its own device state."""
    edges_b, edges_a = np.cumsum([0] + self.nb), np.cumsum([0] + self.na)
    secs = [(self.b[:, edges_b[i]:edges_b[i + 1]], self.a[:, edges_a[i]:edges_a[i + 1]])
            for i in range(len(self.nb))]
    twin = FilterBank(secs, n_inputs=self.n_inputs, mode=self.mode, device=self.device)
    if self._fused:
      twin.set_fused(True)
    if self._time_parallel:
      twin.set_time_parallel(self._time_parallel)
    if self._input_map:
      twin.set_input_map(self._input_map)
    if getattr(self, "_look_check", "call") != "call":
      twin.set_look_check(self._look_check)
    return twin

  def _blocks(self, seq, block=None, post=None):
    """Generator behind the call protocol: pulls ``block`` items, filters them on the GPU and yields
    the results of the block as a list (scalars for a one-channel bank, rows otherwise).  ``post``:
    an elementwise op of :mod:`audiolazy_amd.maps` applied to every output block on the device."""
    scalar_out = self.channels == 1
    block = block_size() if block is None else block
    it = iter(seq)
    scalars = False
    if self.n_inputs == 1:      # scalar items go through np.fromiter: no list of Python floats
      for first in it:
        scalars = not hasattr(first, "__len__")
        it = itertools.chain([first], it)
        break
    while True:
      if scalars:
        x = np.fromiter(itertools.islice(it, block), dtype=np.float64).reshape(-1, 1)
      else:
        chunk = list(itertools.islice(it, block))
        x = np.asarray(chunk, dtype=np.float64).reshape(len(chunk), self.n_inputs)
      if x.shape[0] == 0:
        return
      y = self.process(x, layout="time")
      if post:
        from . import maps
        y = maps.map_block(post, y, device=self.device)
      yield y[:, 0].tolist() if scalar_out else list(y)
