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
