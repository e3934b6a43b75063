"""The algebra behind the dot-product zero-state pass (its HIP form is k_cdot in audiolazy_amd/csrc/alz_scan.hip since round 5):
the NumPy prototype must reproduce the cascade's own chunk end states and the serial outputs."""
import os
import runpy
import sys

import numpy as np


def test_zero_state_end_states_are_dot_products_with_impulse_responses(capsys):
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  argv = sys.argv
  sys.argv = ["cscan_gemm_prototype.py", "8", "8", "128"]
  try:
    g = runpy.run_path(os.path.join(root, "tools", "experiments", "cscan_gemm_prototype.py"))
  finally:
    sys.argv = argv
  capsys.readouterr()
  scale = np.abs(g["Z"]).max()
  assert np.abs(g["E"] - g["Z"]).max() / scale < 1e-12              # the GEMM against the cascade's own zero-state end states
  assert g["worst"] < 1e-6                                          # carried states after S <- M S + z (short chunks of a 50 Hz band: M is close to the identity)
  assert np.abs(g["y_tp"] - g["y_true"]).max() / np.abs(g["y_true"]).max() < 1e-9   # the mode's contract
