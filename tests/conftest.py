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


def _gpu_ready():
  """True when libalzhip.so loads and sees a HIP device (decided once per session)."""
  try:
    from audiolazy_amd import _ffi
    return _ffi.device_count() >= 1
  except Exception:
    return False


def pytest_collection_modifyitems(config, items):
  # plain `pytest` on a machine without a GPU: the gpu-marked tests are skipped, not errors
  # (on a GPU box nothing is skipped -- the product still fails loudly without its library)
  gpu_items = [it for it in items if it.get_closest_marker("gpu")]
  if gpu_items and not _gpu_ready():
    skip = pytest.mark.skip(reason="needs a real MI355X and audiolazy_amd/libalzhip.so")
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
