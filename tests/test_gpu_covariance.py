"""``lag_matrix`` on the GPU (k_lag_matrix behind alz_lag_matrix_dev) and the covariance-method LPC strategies on
top of it, bit for bit against the reference's own results (tests/golden/covariance.json; reference
lazy_analysis.py:315-342, lazy_lpc.py:275-340) and, batched, against the C oracle."""
import ctypes

import numpy as np
import pytest

from conftest import load_golden
from test_covariance_golden import G, block, expected, filt_outcome, same_bits, unhex

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [c for c in G["lag_matrix"] if "phi" in c],
                         ids=lambda c: "%s-%s" % (c["blk"], c["max_lag"]))
def test_lag_matrix_is_the_reference_s(case):
  import audiolazy_amd as al
  blk = block(case["blk"])
  got = al.lag_matrix(blk, case["max_lag"])
  assert isinstance(got, list) and all(isinstance(v, float) for row in got for v in row)
  assert same_bits(got, unhex(case["phi"]))
  # (round 5: a block this small stays on the host, whose sum is the reference's own arithmetic -- lpc._HOST_TERMS;
  # the kernel itself on the same block, whatever its size:)
  from audiolazy_amd.lpc import lag_matrix_frames
  phi = lag_matrix_frames([float(v) for v in blk], len(blk), len(blk) - 1 if case["max_lag"] is None else case["max_lag"])[0]
  assert same_bits(phi.tolist(), unhex(case["phi"]))
  assert "k_lag_matrix" in al.last_kernel()


@pytest.mark.parametrize("family", ["covar", "kcovar"])
def test_covariance_strategies_match_the_reference(family):
  import audiolazy_amd as al
  for case in G[family]:
    strategy = al.lpc[case.get("alias", family)]
    got = filt_outcome(lambda: strategy(block(case["blk"]), case["order"]))
    assert got == expected(case), (family, case["blk"], case["order"], case.get("alias"))


def test_readme_covariance_example():
  """README.rst:360-373: lpc.covar of the periodic block is 1 + .5 z^-2 - .5 z^-4, and its inverse resynthesises."""
  import audiolazy_amd as al
  blk = [-1., 0., 1., 0.] * 50
  analysis = al.lpc.covar(blk, 4)
  close = lambda a, b: len(a) == len(b) and all(abs(x - y) < 1e-12 for x, y in zip(a, b))   # (the README prints rounded values)
  assert close(analysis.numlist, [1, 0, .5, 0, -.5]) and analysis.denlist == [1]
  residual = list(analysis(blk))
  assert close(residual[:10], [-1., 0., .5, 0., 0., 0., 0., 0., 0., 0.])
  assert close(list((1 / analysis)(residual))[:10], [-1., 0., 1., 0., -1., 0., 1., 0., -1., 0.])


@pytest.mark.parametrize("frame_len,hop,max_lag,n", [(128, 64, 10, 300 * 64 + 64), (33, 7, 32, 5000), (512, 512, 16, 512 * 40),
                                                       (16, 1, 0, 600), (64, 64, 63, 64 * 9)])
def test_batched_frames_against_the_oracle(frame_len, hop, max_lag, n):
  import audiolazy_amd as al
  from oracle import oracle
  rng = np.random.default_rng(frame_len * 1000 + max_lag)
  sig = rng.uniform(-1., 1., n)
  phi = al.lag_matrix_frames(sig, frame_len, max_lag, hop=hop)
  frames = (n - frame_len) // hop + 1
  assert phi.shape == (frames, max_lag + 1, max_lag + 1)
  for f in sorted(set([0, 1, frames // 2, frames - 2, frames - 1])):
    assert same_bits(phi[f], oracle.lag_matrix(sig[f * hop:f * hop + frame_len], max_lag)), f
  assert np.array_equal(phi, phi.transpose(0, 2, 1))               # products commute: symmetric to the bit


def test_c_abi_refuses_bad_shapes():
  from audiolazy_amd import _ffi
  L = _ffi.load()
  buf = _ffi.DevBuf(1024, 0)
  out = _ffi.DevBuf(8 * 81, 0)
  assert L.alz_lag_matrix_dev(buf.ptr, 1, 8, 8, 8, out.ptr, 0, None) == _ffi.E_ARG          # max_lag >= frame_len
  assert "Block length should be higher than order" in L.alz_last_error().decode()
  assert L.alz_lag_matrix_dev(None, 1, 8, 8, 2, out.ptr, 0, None) == _ffi.E_ARG
  assert L.alz_lag_matrix_dev(buf.ptr, 1, 8, 8, -1, out.ptr, 0, None) == _ffi.E_ARG
  assert L.alz_lag_matrix_dev(buf.ptr, 0, 8, 8, 2, out.ptr, 0, None) == 0                     # no frames: nothing to do
