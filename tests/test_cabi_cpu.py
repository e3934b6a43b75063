"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a
GPU and exports every symbol include/alz.h declares; the ctypes table covers
the same set; error codes map to the reference's exception types."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
  text = open(os.path.join(ROOT, "include", "alz.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(alz_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
  from audiolazy_amd import _ffi
  if not os.path.exists(_ffi.LIB_PATH):
    import __graft_entry__
    __graft_entry__.build()
  lib = _ffi.load()
  syms = declared_symbols()
  assert len(syms) >= 20
  for name in syms:
    assert hasattr(lib, name), "libalzhip.so does not export %s" % name
  assert sorted(_ffi.SIGNATURES) == syms
  assert lib.alz_version() == 320


def test_status_codes_map_to_reference_exceptions():
  from audiolazy_amd import _ffi
  _ffi.load()
  with pytest.raises(ZeroDivisionError):
    _ffi.check(_ffi.E_ZERO_GAIN)
  with pytest.raises(ValueError):
    _ffi.check(_ffi.E_NONCAUSAL)
  with pytest.raises(_ffi.ParCorError):
    _ffi.check(_ffi.E_PARCOR)
  assert issubclass(_ffi.ParCorError, ZeroDivisionError)   # reference lazy_lpc.py:37
  with pytest.raises(NotImplementedError):
    _ffi.check(_ffi.E_UNSUPPORTED)
  with pytest.raises(MemoryError):
    _ffi.check(_ffi.E_NOMEM)


def test_no_gpu_means_loud_failure_not_fallback():
  import numpy as np
  import audiolazy_amd as alz
  if alz.device_count() > 0:
    pytest.skip("a GPU is visible here")
  with pytest.raises((RuntimeError, MemoryError)):
    alz.FilterBank([([1.0], [1.0, -0.5])], n_inputs=1).process(np.zeros((4, 1)))


def test_product_never_imports_the_oracle():
  pkg = os.path.join(ROOT, "audiolazy_amd")
  for dirpath, _, files in os.walk(pkg):
    for fn in files:
      if fn.endswith((".py", ".hip", ".h", ".cpp")):
        src = open(os.path.join(dirpath, fn)).read()
        for needle in ("import oracle", "from oracle", "alz_oracle", "libalzoracle", "alzo_"):
          assert needle not in src, "%s reaches into the oracle (%s)" % (fn, needle)


def test_memory_to_hist_reference_rules():
  # reference lazy_filters.py:185-195 (+ left padding, lazy_misc.py:132)
  from audiolazy_amd import memory_to_hist
  assert memory_to_hist(None, 2, .5) == [.5, .5]
  assert memory_to_hist([.7], 2, 0.) == [0., .7]
  assert memory_to_hist([1, 2, 3], 2, 0.) == [1, 2]
  assert memory_to_hist(lambda n: [9.] * n, 3, 0.) == [9., 9., 9.]
  assert memory_to_hist(iter([4, 5, 6]), 1, 0.) == [4]


def test_missing_rccl_is_unsupported_not_a_crash(tmp_path):
  """include/alz.h promises ALZ_E_UNSUPPORTED from alz_comm_* when librccl cannot be loaded.  The loader is made
  to refuse every "rccl" name WITHOUT setting dlerror (a preloaded dlopen): round 3 built its message from a
  second dlerror() call, i.e. from NULL."""
  import subprocess
  import sys
  stub = tmp_path / "norccl.c"
  stub.write_text(r'''
#define _GNU_SOURCE
#include <dlfcn.h>
#include <string.h>
void *dlopen(const char *name, int flags) {
  static void *(*real)(const char *, int);
  if (!real) real = (void *(*)(const char *, int))dlsym(RTLD_NEXT, "dlopen");
  if (name && strstr(name, "rccl")) return 0;
  return real(name, flags);
}
''')
  so = tmp_path / "norccl.so"
  subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(stub), "-ldl"])
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  code = (
    "import ctypes, sys\n"
    "L = ctypes.CDLL(%r)\n"
    "L.alz_last_error.restype = ctypes.c_char_p\n"
    "buf = ctypes.create_string_buffer(128)\n"
    "rc = L.alz_comm_unique_id(buf)\n"
    "print(rc, L.alz_last_error().decode())\n" % os.path.join(root, "audiolazy_amd", "libalzhip.so"))
  env = dict(os.environ, LD_PRELOAD=str(so))
  res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
  assert res.returncode == 0, res.stderr[-2000:]
  rc, msg = res.stdout.strip().split(" ", 1)
  assert int(rc) == -7 and "librccl.so not found" in msg, res.stdout
