"""Batched LPC by the autocorrelation method on the GPU.

Host mirror of the reference's ``acorr`` (audiolazy/lazy_analysis.py:277-312),
``levinson_durbin`` (audiolazy/lazy_lpc.py:52-136) and ``lpc.kautocor``
(audiolazy/lazy_lpc.py:229-272) for many frames at once; frames are what
``Stream.blocks(size=frame_len, hop=hop)`` would yield from one signal.

Beside the path: the covariance-method callers of the same file -- ``lag_matrix`` (lazy_analysis.py:315-342, on the
GPU, batched as ``lag_matrix_frames``), ``lpc.covar`` / ``lpc.kcovar`` (lazy_lpc.py:275-340), and the lattice /
stability helpers ``parcor``, ``parcor_stable``, ``lsf``, ``lsf_stable``, ``toeplitz`` (:44-49, :343-487), which are
coefficient algebra on the host like the reference's.
"""
import cmath
import itertools
import numbers
import ctypes

import numpy as np

from . import _ffi
from ._ffi import ParCorError  # noqa: F401


_DevBuf = _ffi.DevBuf


def _frame_count(n_samples, frame_len, hop):
  return 0 if n_samples < frame_len else (n_samples - frame_len) // hop + 1


def _lpc_flags(fused, exact):
  return (_ffi.LPC_FUSED if fused else 0) | (_ffi.LPC_DENSE if exact else 0)


def kautocor_frames(sig, frame_len, order, hop=None, device=0, fused=False, exact=False, out=None):
  """lpc.kautocor on every full frame of ``sig``.

  exact=True runs Levinson-Durbin with the reference's own dense inner products in the reference's
  order (ALZ_LPC_DENSE): coefficients and error are then bit-identical to ``lpc.kautocor`` on every
  frame (the default O(order^2) recursion agrees to ~1e-16, inside the 1e-6 contract, and is faster).

  fused=True opts into fused multiply-adds in the autocorrelation sums (faster: the kernel is bound
  by FP64 issue; lags differ from the reference's by ~1e-16 relative, so the result is no longer
  pinned to the last bit -- the contract is 1e-6).

  out=(coefs, err, status): preallocated CUDA result tensors for the torch path (a caller that analyses block
  after block reuses them instead of allocating three tensors per call).

  sig : 1-D float64 signal (NumPy) or a [F, frame_len] array of frames, or a
        1-D float64 torch CUDA tensor (results are then CUDA tensors).
  Returns (coefs [F, order+1], error [F], status [F]); status is 0 or
  ``_ffi.E_PARCOR`` where the reference would raise ParCorError
  (lazy_lpc.py:132-133).
  """
  L = _ffi.load()
  hop = frame_len if hop is None else hop
  if type(sig).__module__.startswith("torch"):
    import torch
    flat = sig.reshape(-1)
    F = _frame_count(flat.numel(), frame_len, hop)
    if out is not None:
      coefs, err, status = out
      if (tuple(coefs.shape) != (F, order + 1) or tuple(err.shape) != (F,) or tuple(status.shape) != (F,)
          or coefs.dtype != torch.float64 or err.dtype != torch.float64 or status.dtype != torch.int32
          or not (coefs.is_contiguous() and err.is_contiguous() and status.is_contiguous())
          or coefs.device != sig.device or err.device != sig.device or status.device != sig.device):
        raise ValueError("out must be contiguous (coefs [F, order+1] float64, err [F] float64, status [F] int32) on the signal's device")
    else:
      coefs = torch.empty((F, order + 1), dtype=torch.float64, device=sig.device)
      err = torch.empty((F,), dtype=torch.float64, device=sig.device)
      status = torch.empty((F,), dtype=torch.int32, device=sig.device)
    stream = torch.cuda.current_stream(sig.device).cuda_stream
    _ffi.check(L.alz_lpc_kautocor_dev_ex(flat.data_ptr(), F, frame_len, hop, order, coefs.data_ptr(),
                                         err.data_ptr(), status.data_ptr(), _lpc_flags(fused, exact),
                                         sig.device.index or 0, ctypes.c_void_p(stream)))
    return coefs, err, status
  flat = np.ascontiguousarray(sig, dtype=np.float64).reshape(-1)
  F = _frame_count(flat.size, frame_len, hop)
  d_sig = _DevBuf(flat.nbytes, device).upload(flat)
  d_c, d_e, d_s = _DevBuf(F * (order + 1) * 8, device), _DevBuf(F * 8, device), _DevBuf(F * 4, device)
  _ffi.check(L.alz_lpc_kautocor_dev_ex(d_sig.ptr, F, frame_len, hop, order, d_c.ptr, d_e.ptr, d_s.ptr,
                                       _lpc_flags(fused, exact), device, None))
  _ffi.check(L.alz_device_sync(device))
  return (d_c.download((F, order + 1), np.float64), d_e.download((F,), np.float64),
          d_s.download((F,), np.int32))


def acorr_frames(sig, frame_len, max_lag, hop=None, device=0):
  """acorr(blk, max_lag) for every full frame; bit-exact (same summation order
  as lazy_analysis.py:311-312).  Returns r [F, max_lag+1]."""
  L = _ffi.load()
  hop = frame_len if hop is None else hop
  flat = np.ascontiguousarray(sig, dtype=np.float64).reshape(-1)
  F = _frame_count(flat.size, frame_len, hop)
  d_sig = _DevBuf(flat.nbytes, device).upload(flat)
  d_r = _DevBuf(F * (max_lag + 1) * 8, device)
  _ffi.check(L.alz_acorr_dev(d_sig.ptr, F, frame_len, hop, max_lag, d_r.ptr, device, None))
  _ffi.check(L.alz_device_sync(device))
  return d_r.download((F, max_lag + 1), np.float64)


def lag_matrix_frames(sig, frame_len, max_lag, hop=None, device=0):
  """lag_matrix(blk, max_lag) for every full frame: phi [F, max_lag+1, max_lag+1], cell (j, i) the sum of
  ``blk[n - i] * blk[n - j]`` over n = max_lag .. frame_len - 1 in the reference's order (lazy_analysis.py:340-342),
  so the doubles are the reference's."""
  L = _ffi.load()
  hop = frame_len if hop is None else hop
  if max_lag >= frame_len:
    raise ValueError("Block length should be higher than order")
  flat = np.ascontiguousarray(sig, dtype=np.float64).reshape(-1)
  F = _frame_count(flat.size, frame_len, hop)
  P = max_lag + 1
  d_sig = _DevBuf(max(flat.nbytes, 8), device).upload(flat)
  d_phi = _DevBuf(max(F * P * P * 8, 8), device)
  _ffi.check(L.alz_lag_matrix_dev(d_sig.ptr, F, frame_len, hop, max_lag, d_phi.ptr, device, None))
  _ffi.check(L.alz_device_sync(device))
  return d_phi.download((F, P, P), np.float64)


# ---------------------------------------------------------------------------
# the reference's single-block operator surface (lazy_analysis.py:277-312,
# lazy_lpc.py:52-136, 229-272), executed on the GPU one frame at a time
# ---------------------------------------------------------------------------
def _block_fits_engine(blk):
  """The float64 engine takes a block of real floats, Python ints among them only while every product and every
  partial sum of products stays an exact double (then its doubles are the reference's).  A block of nothing but ints
  is integer arithmetic in the reference (doctest lazy_analysis.py:298-306 prints ints), and bools, NumPy integers,
  complex / Fraction / symbolic items never were floats: those run the reference's sums on the host."""
  any_float, biggest = False, 0
  for v in blk:
    if isinstance(v, float):          # Python floats and numpy.float64 (a float subclass); numpy.float32 items keep the
      any_float = True                # reference's per-item arithmetic on the host (their products are float32 there)
    elif type(v) is int:
      biggest = max(biggest, abs(v))
    else:
      return False
  return any_float and biggest * biggest * len(blk) < (1 << 53)


# One block is a GPU round trip (allocate, upload, launch, synchronise, download: ~0.3 ms) -- below this many
# multiply-adds the host's own sum, which IS the reference's arithmetic, is faster (round-4 advisor)
_HOST_TERMS = 8192


def _keep_item_type(blk, values):
  """The reference's sums of numpy.float64 items are numpy.float64: keep that when the whole block is."""
  if len(blk) and all(type(v) is np.float64 for v in blk):
    return [np.float64(v) if isinstance(v, float) else v for v in values]
  return values


def _int_runs(blk):
  """all_int(lo, hi): whether blk[lo:hi] holds only ints (prefix counts of the non-ints)."""
  seen = [0]
  for v in blk:
    seen.append(seen[-1] + (type(v) is not int))
  return lambda lo, hi: seen[hi] == seen[lo]


def acorr(blk, max_lag=None):
  """Autocorrelation of a block for lags 0..max_lag (default len(blk) - 1);
  same summation order as the reference, so the doubles are identical."""
  if max_lag is None:
    max_lag = len(blk) - 1
  size = len(blk)
  host_sum = lambda tau: sum(blk[n] * blk[n + tau] for n in range(size - tau))
  if size * (max_lag + 1) <= _HOST_TERMS or not _block_fits_engine(blk):
    return [host_sum(tau) for tau in range(max_lag + 1)]
  lags = acorr_frames([float(v) for v in blk], size, max_lag)[0].tolist()
  all_int = _int_runs(blk)
  for tau in range(min(max_lag + 1, size + 1)):       # a lag whose terms are all int x int is an int in the reference
    if all_int(0, size - tau) and all_int(tau, size):
      lags[tau] = host_sum(tau)
  for tau in range(size + 1, max_lag + 1):
    lags[tau] = 0                                       # (no terms at all: sum() of nothing)
  return _keep_item_type(blk, lags)


def lag_matrix(blk, max_lag=None):
  """The lag (covariance) matrix of a block as a list of lists (reference lazy_analysis.py:315-342): cell (j, i) is
  the sum of ``blk[n - i] * blk[n - j]`` over every n that needs no padding.  Float blocks are summed on the GPU in
  the reference's order (``lag_matrix_frames``)."""
  if max_lag is None:
    max_lag = len(blk) - 1
  elif max_lag >= len(blk):
    raise ValueError("Block length should be higher than order")
  size = len(blk)
  host_sum = lambda i, j: sum(blk[n - i] * blk[n - j] for n in range(max_lag, size))
  if max_lag < 0 or (size - max_lag) * (max_lag + 1) ** 2 <= _HOST_TERMS or not _block_fits_engine(blk):
    return [[host_sum(i, j) for i in range(max_lag + 1)] for j in range(max_lag + 1)]
  phi = lag_matrix_frames([float(v) for v in blk], size, max_lag)[0].tolist()
  all_int = _int_runs(blk)
  for j in range(max_lag + 1):                          # cells whose terms are all int x int are ints in the reference
    for i in range(max_lag + 1):
      if all_int(max_lag - i, size - i) and all_int(max_lag - j, size - j):
        phi[j][i] = host_sum(i, j)
  return [_keep_item_type(blk, row) for row in phi]


def toeplitz(vect):
  """The symmetric Toeplitz matrix (list of lists) with ``vect`` as its first row and column (lazy_lpc.py:44-49)."""
  size = len(vect)
  return [[vect[abs(col - row)] for col in range(size)] for row in range(size)]


def levinson_durbin(acdata, order=None, device=0, exact=True):
  """Solve the Yule-Walker equations for the lag list ``acdata``; returns the FIR
  analysis filter as a ZFilter with the prediction error in ``.error``
  (reference lazy_lpc.py:52-136).  Raises ParCorError like the reference.
  One lag list at a time, so the reference's dense form (bit-identical result) is the default."""
  from .filters import ZFilter
  L = _ffi.load()
  acdata = np.ascontiguousarray([float(v) for v in acdata], dtype=np.float64)
  if order is None:
    order = len(acdata) - 1
  d_r = _DevBuf(acdata.nbytes, device).upload(acdata)
  d_c, d_e, d_s = _DevBuf((order + 1) * 8, device), _DevBuf(8, device), _DevBuf(4, device)
  _ffi.check(L.alz_levinson_dev_ex(d_r.ptr, 1, len(acdata), order, d_c.ptr, d_e.ptr, d_s.ptr,
                                   _ffi.LPC_DENSE if exact else 0, device, None))
  _ffi.check(L.alz_device_sync(device))
  _ffi.check(int(d_s.download((1,), np.int32)[0]))
  filt = ZFilter(d_c.download((order + 1,), np.float64).tolist())
  filt.error = float(d_e.download((1,), np.float64)[0])
  return filt


def _default_order(blk, order):
  """``order`` defaults to ``len(blk) - 1`` (the reference's documented default, lazy_lpc.py:154-155, which is
  what its ``acorr(blk, None)`` delivers)."""
  return len(blk) - 1 if order is None else order


def _kautocor(blk, order=None, device=0):
  """lpc.kautocor: autocorrelation method via Levinson-Durbin (reference
  lazy_lpc.py:229-272).  Returns a ZFilter with ``.error``.  Any order: past 63 the engine runs the
  reference's dense Levinson-Durbin from a workspace in device memory (slow, but ``lpc(blk, order >= 100)``
  is this route in the reference, :176-180)."""
  from .filters import ZFilter
  blk = [float(v) for v in blk]
  order = _default_order(blk, order)
  coefs, err, status = kautocor_frames(blk, len(blk), order, device=device, exact=True)   # one block: bit-identical form
  _ffi.check(int(status[0]))
  filt = ZFilter(coefs[0].tolist())
  filt.error = float(err[0])
  return filt


def _nautocor(blk, order=None, device=0):
  """lpc.nautocor: the autocorrelation normal equations solved with numpy.linalg.pinv (reference
  lazy_lpc.py:188-225).  The lags come from the GPU (bit-exact ``acorr``); the small dense solve
  is the same NumPy call the reference makes, on the host."""
  from .filters import ZFilter
  blk = [float(v) for v in blk]
  order = _default_order(blk, order)
  lags = np.asarray(acorr(blk, order), dtype=np.float64)
  idx = np.abs(np.subtract.outer(np.arange(order), np.arange(order)))
  normal = lags[idx] if order > 0 else np.zeros((0, 0))
  solved = np.dot(np.linalg.pinv(normal), -lags[1:].reshape(-1, 1)) if order > 0 else np.zeros((0, 1))
  coefs = solved[:, 0].tolist()
  filt = ZFilter([1] + coefs)
  filt.error = float(lags[0]) + sum(r * c for r, c in zip(lags[1:].tolist(), coefs))
  return filt


def _autocor(blk, order=None, device=0):
  """lpc.autocor, the reference's default strategy (lazy_lpc.py:140-185): the pseudo-inverse form
  below order 100, Levinson-Durbin above it with the pseudo-inverse as the ParCorError fallback."""
  blk = [float(v) for v in blk]
  order = _default_order(blk, order)
  if order < 100:
    return _nautocor(blk, order, device=device)
  try:
    return _kautocor(blk, order, device=device)
  except ParCorError:
    return _nautocor(blk, order, device=device)


def _covar(blk, order=None, device=0):
  """lpc.covar: the covariance method, its normal equations solved with numpy.linalg.pinv (reference
  lazy_lpc.py:275-294).  The lag matrix comes from the GPU; the dense solve is the NumPy call the reference makes."""
  from .filters import z
  lagm = lag_matrix(blk, order)
  phi = np.array(lagm)
  solved = np.dot(np.linalg.pinv(phi[1:, 1:]), -phi[1:, :1])
  coeffs = solved.T.tolist()[0]
  filt = 1 + sum(ai * z ** -i for i, ai in enumerate(coeffs, 1))
  filt.error = phi[0, 0] + sum(a * c for a, c in zip(lagm[0][1:], coeffs))
  return filt


def _kcovar(blk, order=None, device=0):
  """lpc.kcovar: the covariance statistics solved greedily, one lattice-like stage per coefficient (reference
  lazy_lpc.py:297-340): a basis B_0, B_1, ... of delayed filters orthogonal under the inner product the lag matrix
  defines, the m-th coefficient being minus the projection of z**-m's partner on B_{m-1}.  ``ValueError("Unstable
  filter")`` when a coefficient leaves (-1, 1), ``ZeroDivisionError`` when a basis filter has no energy."""
  from .filters import ZFilter, z
  phi = lag_matrix(blk, order)
  order = len(phi) - 1

  def inner(a, b):
    total = 0
    for i, ai in enumerate(a.numlist):
      for j, bj in enumerate(b.numlist):
        total = total + phi[i][j] * ai * bj
    return total

  analysis = ZFilter(1)
  basis = [z ** -1]
  energy = [inner(basis[0], basis[0])]
  for m in itertools.count(1):
    try:
      k = -inner(analysis, z ** -m) / energy[m - 1]
    except ZeroDivisionError:
      raise ZeroDivisionError("Can't find next coefficient")
    if k >= 1 or k <= -1:
      raise ValueError("Unstable filter")
    analysis += k * basis[m - 1]
    if m >= order:
      analysis.error = inner(analysis, analysis)
      return analysis
    delayed = z ** -(m + 1)
    shares = [inner(delayed, basis[q]) / energy[q] for q in range(m)]
    basis.append(delayed - sum(shares[q] * basis[q] for q in range(m)))
    energy.append(inner(basis[m], basis[m]))


def _monic_fir(fir_filt):
  """The filter with its constant denominator divided out; feedback is a ValueError (lazy_lpc.py:384-388)."""
  den = fir_filt.denominator
  if len(den) != 1:
    raise ValueError("Filter has feedback")
  if den[0] != 1:
    fir_filt = fir_filt / den[0]
  return fir_filt


def parcor(fir_filt):
  """Generator of the partial correlation (reflection) coefficients of a FIR filter, last stage first: Levinson-
  Durbin run backwards (reference lazy_lpc.py:343-395).  ParCorError when a coefficient of magnitude one stops the
  decomposition."""
  from .filters import z
  fir_filt = _monic_fir(fir_filt)
  for m in range(len(fir_filt.numerator) - 1, 0, -1):
    k = fir_filt.numpoly[m]
    yield k
    mirrored = fir_filt(1 / z) * z ** -m
    try:
      fir_filt = (fir_filt - k * mirrored) / (1 - k ** 2)
    except ZeroDivisionError:
      raise ParCorError("Can't find next PARCOR coefficient")
    fir_filt = (fir_filt - fir_filt.numpoly[0]) + 1   # the leading 1 again, whatever the rounding made of it


def parcor_stable(filt):
  """True when every reflection coefficient of the filter's denominator lies inside the unit circle (reference
  lazy_lpc.py:398-425); a coefficient ON it (critical stability, or a ParCorError) counts as unstable."""
  from .filters import ZFilter
  try:
    return all(abs(k) < 1 for k in parcor(ZFilter(filt.denpoly)))
  except ParCorError:
    return False


def lsf(fir_filt):
  """Line spectral frequencies of a FIR filter in rad/sample (reference lazy_lpc.py:428-457): the root angles of
  the palindromic P = A + z**-1 A(1/z) and the antipalindromic Q = A - z**-1 A(1/z), each sorted, then interleaved
  starting with the family that holds the lowest angle.  (The reference takes the angles with its elementwise
  ``phase``, which NumPy 2 no longer lets it apply to an array; this is the same cmath.phase per root.)"""
  from .filters import ZFilter, z
  fir_filt = _monic_fir(fir_filt)
  rev_filt = ZFilter(fir_filt.numerator[::-1]) * z ** -1
  angles = []
  for family in (fir_filt + rev_filt, fir_filt - rev_filt):
    roots = np.roots(family.numerator[::-1])
    angles.append(sorted(np.array([cmath.phase(r) for r in roots])))
  out = ()
  for pair in zip(*sorted(angles)):
    out = out + pair
  return out


def lsf_stable(filt):
  """True when the line spectral frequencies of the filter's denominator strictly alternate between the two
  families (reference lazy_lpc.py:460-487); equal neighbours count as unstable."""
  from .filters import ZFilter
  lsf_data = lsf(ZFilter(filt.denpoly))
  return all(lsf_data[i] < lsf_data[i + 1] for i in range(len(lsf_data) - 1))


def _make_lpc():
  from .strategy import StrategyDict
  sd = StrategyDict("lpc")
  sd.strategy("autocor", "acorr", "autocorrelation", "auto_correlation")(_autocor)
  sd.strategy("nautocor", "nacorr", "nautocorrelation", "nauto_correlation")(_nautocor)
  sd.strategy("kautocor", "kacorr", "kautocorrelation", "kauto_correlation")(_kautocor)
  sd.strategy("covar", "cov", "covariance", "ncovar", "ncov", "ncovariance")(_covar)
  sd.strategy("kcovar", "kcov", "kcovariance")(_kcovar)
  return sd


# The autocorrelation-method strategies, with the reference's default (``lpc(blk, order)`` is
# ``lpc.autocor``).  ``kautocor`` is the one BASELINE/SURVEY put on the hot path (batched:
# ``kautocor_frames``); the covariance methods (``covar`` / ``kcovar``) take their lag matrix from the GPU and do
# the reference's small solves on the host.
lpc = _make_lpc()
