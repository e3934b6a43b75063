"""Elementwise stages around the filter, on the device (include/alz.h: alz_map_dev).

The reference writes ``abs(sig)``, ``sig * gain``, ``a + b``, ``clip(sig)``, ``... ** .5`` as lazy
per-sample Stream expressions (audiolazy/lazy_stream.py:47-71, lazy_analysis.py:440-520, 619-647).
For blocks that already live in arrays -- thousands of channels -- these functions run the same
IEEE operations on the GPU: results are bit-identical to the per-sample Python expressions, except
``square`` (``x * x``, which is not libm's ``pow(x, 2.0)`` in the last bit of ~0.1 % of samples and
is therefore never used implicitly).

Arrays in, arrays out: NumPy arrays are staged through the device, contiguous float64 torch CUDA
tensors are processed where they are (on torch's current stream).
"""
import ctypes

import numpy as np

from . import _ffi

__all__ = ["map_block", "abs_block", "neg_block", "sqrt_block", "square_block", "clip_block", "scale_block",
           "add_blocks", "sub_blocks", "mul_blocks", "div_blocks"]


def _is_torch(x):
  return type(x).__module__.startswith("torch")


def _raise_flags(flags):
  if flags & _ffi.MAP_ZERODIV:
    raise ZeroDivisionError("float division by zero")
  if flags & _ffi.MAP_DOMAIN:
    raise NotImplementedError("root of a negative sample: the reference yields a complex number there, "
                              "the engine computes in float64")


def map_block(op, x, other=None, p0=0., p1=0., out=None, device=0):
  """``out[i] = op(x[i], ...)`` for one of the ops of ``_ffi.MAP_OPS`` (see include/alz.h):
  unary ``abs neg sqrt square``, with a scalar ``mul add sub rsub div rdiv`` (``p0``), clipping
  ``clip clip_low clip_high`` (``p0`` = low, ``p1`` = high), two-operand ``add2 sub2 mul2 div2``
  (``other``).  Raises what the per-sample Python expression would raise (ZeroDivisionError)."""
  L = _ffi.load()
  code = _ffi.MAP_OPS[op]
  if op == "div" and p0 == 0:
    raise ZeroDivisionError("float division by zero")
  binary = code >= 20
  if binary and other is None:
    raise ValueError("%s needs a second block" % op)
  if _is_torch(x):
    import torch
    for t in (x, other) if binary else (x,):
      if not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous():
        raise ValueError("torch blocks must be contiguous float64 CUDA tensors")
    if binary and tuple(other.shape) != tuple(x.shape):
      raise ValueError("blocks differ in shape")
    if out is None:
      res = torch.empty_like(x)
    else:       # the kernel writes x.numel() doubles through out.data_ptr(): it must be exactly that buffer
      if (not getattr(out, "is_cuda", False) or out.dtype != torch.float64 or not out.is_contiguous()
          or tuple(out.shape) != tuple(x.shape) or out.device != x.device):
        raise ValueError("out must be a contiguous float64 CUDA tensor of the input's shape on the input's device")
      res = out
    flags = torch.zeros((1,), dtype=torch.int32, device=x.device)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    _ffi.check(L.alz_map_dev(code, x.data_ptr(), other.data_ptr() if binary else None, float(p0), float(p1),
                             x.numel(), res.data_ptr(), flags.data_ptr(), x.device.index or 0,
                             ctypes.c_void_p(stream)))
    if op in ("sqrt", "rdiv", "div2"):
      _raise_flags(int(flags.item()))
    return res
  x = np.ascontiguousarray(x, dtype=np.float64)
  if binary:
    other = np.ascontiguousarray(other, dtype=np.float64)
    if other.shape != x.shape:
      raise ValueError("blocks differ in shape")
  if out is not None and (not isinstance(out, np.ndarray) or out.shape != x.shape or out.dtype != np.float64):
    raise ValueError("out must be a float64 ndarray of the input's shape")
  if x.size == 0:
    return np.empty(x.shape) if out is None else out
  d_x = _ffi.DevBuf(x.nbytes, device).upload(x)
  d_y = _ffi.DevBuf(other.nbytes, device).upload(other) if binary else None
  d_f = _ffi.DevBuf(4, device).upload(np.zeros(1, dtype=np.int32))
  _ffi.check(L.alz_map_dev(code, d_x.ptr, d_y.ptr if binary else None, float(p0), float(p1), x.size, d_x.ptr,
                           d_f.ptr, device, None))
  _ffi.check(L.alz_device_sync(device))
  _raise_flags(int(d_f.download((1,), np.int32)[0]))
  res = d_x.download(x.shape, np.float64)
  if out is not None:
    out[...] = res
    return out
  return res


def abs_block(x, **kw):
  """``abs(Stream(x))`` (lazy_stream.py:60-64)."""
  return map_block("abs", x, **kw)


def neg_block(x, **kw):
  """``-Stream(x)``."""
  return map_block("neg", x, **kw)


def sqrt_block(x, **kw):
  """``Stream(x) ** .5`` for non-negative samples (the root of envelope.rms, lazy_analysis.py:465)."""
  return map_block("sqrt", x, **kw)


def square_block(x, **kw):
  """``x * x`` -- NOT bit-identical to the reference's ``Stream(x) ** 2`` (libm pow): the correctly
  rounded product differs from glibc's pow(x, 2.0) in the last bit of roughly 0.1 % of samples."""
  return map_block("square", x, **kw)


def scale_block(x, gain, **kw):
  """``Stream(x) * gain`` (== ``gain * Stream(x)``)."""
  return map_block("mul", x, p0=gain, **kw)


def clip_block(x, low=-1., high=1., **kw):
  """``clip(x, low, high)`` with the reference's rules (lazy_analysis.py:619-647): either limit may
  be None; ``high < low`` is a ValueError; a NaN passes the two-sided form and becomes the limit in
  the one-sided ones."""
  if low is None and high is None:
    return x.clone() if _is_torch(x) else np.array(x, dtype=np.float64)
  if low is None:
    return map_block("clip_high", x, p1=high, **kw)
  if high is None:
    return map_block("clip_low", x, p0=low, **kw)
  if high < low:
    raise ValueError("Higher clipping limit is smaller than lower one")
  return map_block("clip", x, p0=low, p1=high, **kw)


def add_blocks(x, y, **kw):
  return map_block("add2", x, other=y, **kw)


def sub_blocks(x, y, **kw):
  return map_block("sub2", x, other=y, **kw)


def mul_blocks(x, y, **kw):
  return map_block("mul2", x, other=y, **kw)


def div_blocks(x, y, **kw):
  return map_block("div2", x, other=y, **kw)
