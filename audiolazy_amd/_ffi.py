"""ctypes binding of libalzhip.so (C ABI declared in include/alz.h).

There is no CPU execution path behind this module: if the HIP library is
missing or fails to load, importing a compute entry point raises ImportError
loudly instead of falling back to anything else.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ALZ_LIBRARY") or os.path.join(_HERE, "libalzhip.so")   # ALZ_LIBRARY: A/B builds of the same ABI

# status codes (include/alz.h)
OK = 0
TV_NEGATED = 1
E_ARG, E_NONCAUSAL, E_ZERO_GAIN, E_PARCOR, E_HIP, E_NOMEM, E_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7
TIME_MAJOR, CHAN_MAJOR = 0, 1
MAP_OPS = {"abs": 1, "neg": 2, "sqrt": 3, "square": 4, "mul": 5, "add": 6, "sub": 7, "rsub": 8, "div": 9, "rdiv": 10,
           "clip": 11, "clip_high": 12, "clip_low": 13, "add2": 20, "sub2": 21, "mul2": 22, "div2": 23}
MAP_ZERODIV, MAP_DOMAIN = 1, 2
LPC_FUSED = 1
LPC_DENSE = 2
BANK_DIAGONAL, BANK_OUTER = 0, 1

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_u64 = ctypes.c_uint64
_int = ctypes.c_int

# every symbol include/alz.h declares: name -> (restype, argtypes)
SIGNATURES = {
  "alz_version": (_int, []),
  "alz_last_error": (ctypes.c_char_p, []),
  "alz_last_kernel": (ctypes.c_char_p, []),
  "alz_device_count": (_int, [_ip]),
  "alz_malloc": (_int, [_int, _u64, ctypes.POINTER(_vp)]),
  "alz_free": (_int, [_int, _vp]),
  "alz_memcpy_h2d": (_int, [_int, _vp, _vp, _u64]),
  "alz_memcpy_d2h": (_int, [_int, _vp, _vp, _u64]),
  "alz_device_sync": (_int, [_int]),
  "alz_bank_create": (_int, [_i64, _i64, _int, _int, _ip, _ip, _dp, _dp, _int, ctypes.POINTER(_vp)]),
  "alz_bank_destroy": (_int, [_vp]),
  "alz_bank_channels": (_int, [_vp, ctypes.POINTER(_i64)]),
  "alz_bank_reset": (_int, [_vp, ctypes.c_double]),
  "alz_bank_set_state": (_int, [_vp, _dp, _dp]),
  "alz_bank_get_state": (_int, [_vp, _dp, _dp]),
  "alz_bank_process_dev": (_int, [_vp, _vp, _vp, _i64, _int, _i64, _i64, _vp]),
  "alz_bank_process_host": (_int, [_vp, _dp, _dp, _i64, _int, _i64, _i64]),
  "alz_bank_sync": (_int, [_vp]),
  "alz_bank_set_fused": (_int, [_vp, _int]),
  "alz_bank_set_time_parallel": (_int, [_vp, _i64]),
  "alz_bank_set_input_map": (_int, [_vp, _int]),
  "alz_bank_set_look_check": (_int, [_vp, _int]),
  "alz_bank_look_stats": (_int, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                                 ctypes.POINTER(ctypes.c_uint)]),
  "alz_map_dev": (_int, [_int, _vp, _vp, ctypes.c_double, ctypes.c_double, _i64, _vp, _vp, _int, _vp]),
  "alz_bank_last_kernel": (ctypes.c_char_p, [_vp]),
  "alz_lpc_kautocor_dev": (_int, [_vp, _i64, _int, _i64, _int, _vp, _vp, _vp, _int, _vp]),
  "alz_lpc_kautocor_dev_ex": (_int, [_vp, _i64, _int, _i64, _int, _vp, _vp, _vp, _int, _int, _vp]),
  "alz_levinson_dev": (_int, [_vp, _i64, _int, _int, _vp, _vp, _vp, _int, _vp]),
  "alz_levinson_dev_ex": (_int, [_vp, _i64, _int, _int, _vp, _vp, _vp, _int, _int, _vp]),
  "alz_acorr_dev": (_int, [_vp, _i64, _int, _i64, _int, _vp, _int, _vp]),
  "alz_lag_matrix_dev": (_int, [_vp, _i64, _int, _i64, _int, _vp, _int, _vp]),
  "alz_mix_dev": (_int, [_vp, _i64, _i64, _i64, _int, _i64, _i64, _vp, _int, _vp]),
  "alz_mix_tracks_dev": (_int, [_int, _vp, _vp, _vp, ctypes.c_double, _i64, _vp, _int, _vp]),
  "alz_pcm_decode_dev": (_int, [_vp, _int, _int, _i64, _vp, _int, _vp]),
  "alz_pcm_encode_dev": (_int, [_vp, _i64, _int, _int, _vp, _vp, _int, _vp]),
  "alz_comm_unique_id": (_int, [_vp]),
  "alz_comm_create": (_int, [_int, _int, _int, _vp, ctypes.POINTER(_vp)]),
  "alz_comm_destroy": (_int, [_vp]),
  "alz_comm_gather": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
  "alz_comm_sum": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
  "alz_tv_process_dev": (_int, [_int, _vp, _int, _vp, _i64, _vp, _vp, _i64, _int, _i64, _i64, _vp, _vp,
                                ctypes.c_double, _int, _vp]),
}


# wait sites of the one-pass time-parallel kernel (csrc/alz_look.hip W_*: bit k of alz_bank_look_stats' last_sites)
LOOK_WAIT_SITES = ("?", "LOAD/stored", "HELP/prepared", "HELP/replayed", "HELP/state-read", "CHAIN/published-states",
                   "CHAIN/replayed", "CHAIN/summed", "CHAIN/prepared", "REPLAY/start-state", "LOAD/all-stored", "CHAIN/neighbour-start-state",
                   "CHUNK-BOUNDARY-CHECK-FAILED")


class TvTap(ctypes.Structure):
  """alz_tv_tap_t (include/alz.h)."""
  _fields_ = [("value", ctypes.c_double), ("series_dev", ctypes.c_void_p),
              ("stride_n", ctypes.c_int64), ("stride_c", ctypes.c_int64), ("flags", ctypes.c_int64)]


class ParCorError(ZeroDivisionError):
  """Error when trying to find the partial correlation coefficients
  (reference audiolazy/lazy_lpc.py:37-41)."""


_lib = None


def _pin_hip_runtime():
  """One HIP runtime per process.

  PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so
  (same SONAMEs as /opt/rocm's).  Whichever copy is loaded first serves every
  later DT_NEEDED lookup, and a process that ends up with both (libalzhip.so
  initialising /opt/rocm's copy, torch its own) sees "No HIP GPUs are
  available" from the second one.  So before libalzhip.so is loaded, make
  torch's copy the resident one when torch is installed; device pointers and
  streams handed over from torch tensors then belong to the same runtime.
  """
  import importlib.util
  import sys
  if "torch" in sys.modules:
    return
  try:
    spec = importlib.util.find_spec("torch")
  except (ImportError, ValueError):
    spec = None
  if spec is None or not spec.origin:
    return
  path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
  if os.path.exists(path):
    try:
      ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except OSError:
      pass


def load():
  """Load libalzhip.so and bind every declared symbol (no GPU needed for this)."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise ImportError(
      "audiolazy_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
      "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
  _pin_hip_runtime()
  try:
    L = ctypes.CDLL(LIB_PATH)
  except OSError as exc:  # e.g. no ROCm runtime on this machine
    raise ImportError("audiolazy_amd: cannot load %s: %s" % (LIB_PATH, exc))
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(L, name)  # AttributeError here == header/library mismatch
    fn.restype = res
    fn.argtypes = args
  _lib = L
  return L


def last_error():
  msg = load().alz_last_error()
  return msg.decode("utf-8", "replace") if msg else ""


def last_kernel():
  """Kernel(s) the last handle-less call of this thread launched (alz_last_kernel): lpc / acorr / levinson /
  time-varying blocks.  Banks report theirs through ``FilterBank.last_kernel``."""
  name = load().alz_last_kernel()
  return name.decode("utf-8", "replace") if name else ""


def check(rc):
  """Map a C status to the exception type the reference raises."""
  if rc == OK:
    return
  msg = last_error()
  if rc == E_ZERO_GAIN:
    raise ZeroDivisionError(msg or "Invalid filter gain")   # lazy_filters.py:177-178
  if rc == E_NONCAUSAL:
    raise ValueError(msg or "Non-causal filter")             # lazy_filters.py:165-168
  if rc == E_PARCOR:
    raise ParCorError(msg or "Can't find next PARCOR coefficient")  # lazy_lpc.py:132-133
  if rc == E_ARG:
    raise ValueError(msg)
  if rc == E_NOMEM:
    raise MemoryError(msg)
  if rc == E_UNSUPPORTED:
    raise NotImplementedError(msg)
  raise RuntimeError("libalzhip: %s (status %d)" % (msg, rc))


def device_count():
  n = _int(0)
  rc = load().alz_device_count(ctypes.byref(n))
  return n.value if rc == OK else 0


def require_gpu():
  if device_count() < 1:
    raise RuntimeError("audiolazy_amd: no HIP device visible; this engine has no CPU path (%s)"
                       % last_error())


class DevBuf(object):
  """Device allocation owned through the C ABI (no torch needed)."""

  def __init__(self, nbytes, device=0):
    self.device, self.nbytes = device, int(nbytes)
    self.ptr = ctypes.c_void_p()
    check(load().alz_malloc(device, max(self.nbytes, 8), ctypes.byref(self.ptr)))

  def upload(self, arr):
    arr = np.ascontiguousarray(arr)
    check(load().alz_memcpy_h2d(self.device, self.ptr, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))
    return self

  def download(self, shape, dtype):
    out = np.empty(shape, dtype=dtype)
    check(load().alz_memcpy_d2h(self.device, out.ctypes.data_as(ctypes.c_void_p), self.ptr, out.nbytes))
    return out

  def __del__(self):
    p, self.ptr = getattr(self, "ptr", None), None
    if p:
      try:
        load().alz_free(self.device, p)
      except Exception:
        pass
