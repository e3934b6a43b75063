"""Sample formats either side of the filter path: ``WavStream`` in, ``chunks`` out.

Host mirror of the reference's ingest / egress helpers -- ``WavStream`` (audiolazy/lazy_wav.py
:31-130: a wave file as a Stream of samples scaled to [-1, 1), or the stored integers with
``keep``) and the ``chunks`` StrategyDict (audiolazy/lazy_io.py:44-128: a stream packed into
``size``-item byte strings of one struct format).  File parsing and laziness stay on the host; the
per-sample conversions run on the GPU through ``alz_pcm_decode_dev`` / ``alz_pcm_encode_dev``
and are bit-identical to the reference's (division by a power of two; struct's own rounding for
``"f"``).  ``decode_pcm`` / ``encode_pcm`` are the array-level entry points: interleaved frames
are exactly the engine's time-major ``[N, C]`` block layout, so a decoded file can go straight
into ``FilterBank.process``.
"""
import ctypes
import itertools
import numbers
import struct
import wave

import numpy as np

from . import _ffi
from .strategy import StrategyDict
from .stream import Stream, blocks

__all__ = ["WavStream", "chunks", "decode_pcm", "encode_pcm"]

_WIDTH = {"b": 1, "B": 1, "h": 2, "H": 2, "i": 4, "I": 4, "l": 4, "L": 4, "f": 4, "d": 8}
_INT_RANGE = {"b": (-128, 127), "B": (0, 255), "h": (-32768, 32767), "H": (0, 65535),
              "i": (-2 ** 31, 2 ** 31 - 1), "I": (0, 2 ** 32 - 1),
              "l": (-2 ** 31, 2 ** 31 - 1), "L": (0, 2 ** 32 - 1)}
NOT_INTEGER, RANGE, FLOAT_OVERFLOW = 1, 2, 4   # include/alz.h ALZ_PCM_*


def _is_torch(x):
  return type(x).__module__.startswith("torch")


def _big_endian(byte_order):
  if byte_order in (None, "", "@", "=", "<"):
    return 0      # this engine's hosts are little-endian
  if byte_order in (">", "!"):
    return 1
  raise struct.error("bad char in struct format")


def decode_pcm(raw, bits, keep=False, device=0):
  """Little-endian PCM -> float64, the conversion WavStream applies per sample
  (lazy_wav.py:58-61, 110-128).

  raw : bytes-like / uint8 NumPy array (staged through the device; returns a NumPy array) or a
        contiguous uint8 torch CUDA tensor (returns a CUDA tensor).  Samples stay in file order.
  """
  L = _ffi.load()
  if bits not in (8, 16, 24, 32):
    raise NotImplementedError("bits per sample must be 8, 16, 24 or 32")
  width = bits // 8
  if _is_torch(raw):
    import torch
    if not raw.is_cuda or raw.dtype != torch.uint8 or not raw.is_contiguous():
      raise ValueError("torch input must be a contiguous uint8 CUDA tensor")
    n = raw.numel() // width
    out = torch.empty((n,), dtype=torch.float64, device=raw.device)
    stream = torch.cuda.current_stream(raw.device).cuda_stream
    _ffi.check(L.alz_pcm_decode_dev(raw.data_ptr(), bits, 1 if keep else 0, n, out.data_ptr(),
                                    raw.device.index or 0, ctypes.c_void_p(stream)))
    return out
  buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
  n = buf.size // width
  if n == 0:
    return np.empty((0,))
  d_raw = _ffi.DevBuf(buf.size, device).upload(buf)
  d_out = _ffi.DevBuf(n * 8, device)
  _ffi.check(L.alz_pcm_decode_dev(d_raw.ptr, bits, 1 if keep else 0, n, d_out.ptr, device, None))
  _ffi.check(L.alz_device_sync(device))
  return d_out.download((n,), np.float64)


def _raise_flags(flags, dfmt, strict_float=True, as_array=False):
  if flags & NOT_INTEGER:
    if as_array:
      raise TypeError("integer argument expected, got float")
    raise struct.error("required argument is not an integer")
  if flags & RANGE:
    lo, hi = _INT_RANGE[dfmt]
    if as_array:
      raise OverflowError("value out of range for array typecode %r" % dfmt)
    raise struct.error("%r format requires %d <= number <= %d" % (dfmt, lo, hi))
  if flags & FLOAT_OVERFLOW and strict_float:
    raise OverflowError("float too large to pack with f format")


def encode_pcm(x, dfmt="f", byte_order=None, device=0, _as_array=False):
  """float64 items -> packed bytes in struct format ``dfmt``, what ``struct.pack`` of the same
  items gives (lazy_io.py:89-94).  Integer formats need integer-valued in-range items
  (struct.error otherwise); ``"f"`` overflow raises OverflowError like struct does.

  x : float64 NumPy array / sequence (returns ``bytes``) or a contiguous float64 torch CUDA tensor
      (returns a uint8 CUDA tensor).
  """
  L = _ffi.load()
  if dfmt not in _WIDTH:
    raise NotImplementedError("dfmt must be one of %s" % " ".join(sorted(_WIDTH)))
  width, big = _WIDTH[dfmt], _big_endian(byte_order)
  if _is_torch(x):
    import torch
    if not x.is_cuda or x.dtype != torch.float64 or not x.is_contiguous():
      raise ValueError("torch input must be a contiguous float64 CUDA tensor")
    n = x.numel()
    out = torch.empty((n * width,), dtype=torch.uint8, device=x.device)
    flags = torch.zeros((1,), dtype=torch.int32, device=x.device)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    _ffi.check(L.alz_pcm_encode_dev(x.data_ptr(), n, ord(dfmt), big, out.data_ptr(), flags.data_ptr(),
                                    x.device.index or 0, ctypes.c_void_p(stream)))
    _raise_flags(int(flags.item()), dfmt, strict_float=not _as_array, as_array=_as_array)
    return out
  arr = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
  if arr.size == 0:
    return b""
  d_in = _ffi.DevBuf(arr.nbytes, device).upload(arr)
  d_out = _ffi.DevBuf(arr.size * width, device)
  d_flags = _ffi.DevBuf(4, device).upload(np.zeros(1, dtype=np.int32))
  _ffi.check(L.alz_pcm_encode_dev(d_in.ptr, arr.size, ord(dfmt), big, d_out.ptr, d_flags.ptr, device, None))
  _ffi.check(L.alz_device_sync(device))
  _raise_flags(int(d_flags.download((1,), np.int32)[0]), dfmt, strict_float=not _as_array, as_array=_as_array)
  return d_out.download((arr.size * width,), np.uint8).tobytes()


# ---------------------------------------------------------------------------
# chunks (reference lazy_io.py:44-128)
# ---------------------------------------------------------------------------
chunks = StrategyDict("chunks")
chunks.size = 2048   # samples; the default chunk size, changeable like the reference's chunks.size


def _pack_blocks(seq, size, dfmt, byte_order, padval, lookahead, as_array, device):
  if size is None:
    size = chunks.size
  if dfmt not in _WIDTH:
    raise NotImplementedError("dfmt must be one of %s" % " ".join(sorted(_WIDTH)))
  integer = dfmt in _INT_RANGE
  nbytes = size * _WIDTH[dfmt]
  it = iter(blocks(seq, size, padval=padval))
  while True:
    group = [list(blk) for blk in itertools.islice(it, max(1, int(lookahead)))]
    if not group:
      return
    flat = [v for blk in group for v in blk]
    if integer and not all(isinstance(v, numbers.Integral) for v in flat):
      # struct / array refuse floats in integer formats, integer-valued or not
      if as_array:
        raise TypeError("integer argument expected, got float")
      raise struct.error("required argument is not an integer")
    data = encode_pcm(np.asarray(flat, dtype=np.float64), dfmt, byte_order, device=device, _as_array=as_array)
    for k in range(len(group)):
      yield data[k * nbytes:(k + 1) * nbytes]


@chunks.strategy("struct")
def chunks(seq, size=None, dfmt="f", byte_order=None, padval=0., lookahead=1, device=0):
  """``size``-item byte strings of ``seq`` packed as struct format ``dfmt`` (reference
  lazy_io.py:48-94); the last chunk is padded with ``padval``.  ``byte_order``: None (native),
  "<" or ">".  ``lookahead`` > 1 converts that many chunks per launch (the source is then read
  that far ahead of the consumer); 1 keeps the reference's consumption pattern."""
  return _pack_blocks(seq, size, dfmt, byte_order, padval, lookahead, False, device)


@chunks.strategy("array")
def chunks(seq, size=None, dfmt="f", byte_order=None, padval=0., lookahead=1, device=0):
  """Same chunks with the array module's conventions (reference lazy_io.py:97-128): native byte
  order whatever ``byte_order`` says, ``"f"`` overflow becomes inf instead of raising, integer
  problems surface as TypeError / OverflowError."""
  return _pack_blocks(seq, size, dfmt, None, padval, lookahead, True, device)


chunks.default = chunks.struct


# ---------------------------------------------------------------------------
# WavStream (reference lazy_wav.py:31-130)
# ---------------------------------------------------------------------------
class WavStream(Stream):
  """A Stream of the samples of a wave file, with ``rate``, ``channels`` and ``bits``.

  Like the reference, multichannel data stays serialized (one sample per channel for each time
  instant, in turn): ``blocks(channels)`` regroups the frames.  ``keep=True`` yields the stored
  integers (8-bit files: 0..255); the default scales to [-1, 1).  The file is read lazily,
  ``block_frames`` frames at a time, each block converted on the GPU.  :meth:`array` hands the
  rest of the file over as one ``[frames, channels]`` time-major block instead.
  """

  def __init__(self, wave_file, keep=False, block_frames=1 << 16, device=0):
    self._file = wave.open(wave_file, "rb")
    self.rate = self._file.getframerate()
    self.channels = self._file.getnchannels()
    self.bits = 8 * self._file.getsampwidth()
    if self.bits not in (8, 16, 24, 32):
      raise NotImplementedError("bits per sample must be 8, 16, 24 or 32")
    self.keep, self.device = bool(keep), device
    self._block_frames = int(block_frames)

    def data_generator():
      try:
        while True:
          raw = self._file.readframes(self._block_frames)
          if not raw:
            break
          vals = decode_pcm(raw, self.bits, keep=self.keep, device=self.device)
          if self.keep:
            for v in vals.tolist():
              yield int(v)
          else:
            for v in vals.tolist():
              yield v
      finally:
        self._file.close()

    super(WavStream, self).__init__(data_generator())

  def array(self, as_torch=False):
    """Everything not yet read, as one float64 ``[frames, channels]`` block (time-major: the
    layout ``FilterBank.process`` takes).  ``as_torch`` keeps it on the device."""
    raw = self._file.readframes(self._file.getnframes())
    self._file.close()
    if as_torch:
      import torch
      dev = torch.device("cuda", self.device)
      host = torch.frombuffer(bytearray(raw), dtype=torch.uint8) if raw else torch.empty(0, dtype=torch.uint8)
      vals = decode_pcm(host.to(dev), self.bits, keep=self.keep, device=self.device)
      return vals.reshape(-1, self.channels)
    return decode_pcm(raw, self.bits, keep=self.keep, device=self.device).reshape(-1, self.channels)
