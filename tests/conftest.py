"""pytest configuration: the `gpu` marker and shared golden-fixture helpers."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_state():
  """("ready" | "no-gpu" | "broken", detail), decided once per session.

  ready   libalzhip.so loads and sees a HIP device;
  no-gpu  this machine has no GPU (no /dev/kfd, or the library loads and counts 0 devices): the
          gpu-marked tests are skipped;
  broken  there IS a GPU (or ALZ_REQUIRE_GPU=1 says there must be one) but the library is missing or
          does not load: that is a failure, never a skip -- a broken build must not turn 900 tests green.
  """
  must = os.environ.get("ALZ_REQUIRE_GPU", "") not in ("", "0")
  has_kfd = os.path.exists("/dev/kfd")
  try:
    from audiolazy_amd import _ffi
    _ffi.load()
  except Exception as exc:        # missing .so, unresolved symbol, no ROCm runtime
    if has_kfd or must:
      return "broken", "audiolazy_amd/libalzhip.so does not load on a machine with a GPU: %s" % exc
    return "no-gpu", "no GPU here and no usable libalzhip.so (%s)" % exc
  n = _ffi.device_count()
  if n >= 1:
    return "ready", "%d device(s)" % n
  if must:
    return "broken", "ALZ_REQUIRE_GPU is set but libalzhip.so sees no HIP device (%s)" % _ffi.last_error()
  return "no-gpu", "libalzhip.so loads, no HIP device visible"


def pytest_collection_modifyitems(config, items):
  gpu_items = [it for it in items if it.get_closest_marker("gpu")]
  if not gpu_items:
    return
  state, detail = _gpu_state()
  if state == "broken":
    raise pytest.UsageError("gpu-marked tests were selected but the HIP library is unusable: " + detail)
  if state == "no-gpu":
    skip = pytest.mark.skip(reason="needs a real MI355X (" + detail + ")")
    for it in gpu_items:
      it.add_marker(skip)


def unhex(v):
  """Inverse of gen_golden.hx: nested lists of float.hex() strings -> floats."""
  if isinstance(v, list):
    return [unhex(i) for i in v]
  if isinstance(v, str):
    return float.fromhex(v)
  return v


def load_golden(name):
  with open(os.path.join(GOLDEN, name)) as f:
    return json.load(f)


@pytest.fixture(scope="session")
def golden():
  return load_golden
