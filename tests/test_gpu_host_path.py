"""The host-buffer entry point (alz_bank_process_host) on blocks large enough for its pinned,
pipelined path: copy-in, kernels and copy-out overlap over chunks of the time axis, and the result
must be the same doubles as one pass over the whole block (state carried from chunk to chunk)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def alz():
  import audiolazy_amd
  audiolazy_amd.load_library()
  assert audiolazy_amd.device_count() >= 1, "no HIP device: the engine has no CPU path"
  return audiolazy_amd


def same_bits(a, b):
  return a.shape == b.shape and bool(np.array_equal(a.view(np.uint64), b.view(np.uint64)))


@pytest.mark.parametrize("layout", ["time", "chan"])
def test_large_numpy_block_through_the_pipelined_host_path(alz, layout):
  from oracle import oracle
  import bench
  C, N = 512, 40000 + 6            # 164 MB in + 164 MB out: several chunks, ragged last one
  b, a = bench.resonator_coefs(C)
  x = np.random.default_rng(21).uniform(-1, 1, (N, C) if layout == "time" else (C, N))
  bank = alz.FilterBank([(b, a)], n_inputs=C)
  bank.reset()
  y = bank.process(x, layout=layout)
  assert same_bits(y, oracle.bank([3], [3], b, a, x, layout=layout))
  # and the stream continues across calls
  x2 = np.random.default_rng(22).uniform(-1, 1, (N, C) if layout == "time" else (C, N))
  y2 = bank.process(x2, layout=layout)
  ax = 0 if layout == "time" else 1
  ref = oracle.bank([3], [3], b, a, np.concatenate([x, x2], axis=ax), layout=layout)
  assert same_bits(y2, ref[N:] if layout == "time" else ref[:, N:])


def test_outer_bank_and_cascade_through_the_pipelined_host_path(alz):
  from oracle import oracle
  s, Hz = alz.sHz(48000)
  B, S, N = 16, 64, 16384
  fcs = [f * Hz for f in alz.erb_space(100., 8000., B)]
  bank = alz.gammatone_bank(fcs, S, strategy="slaney", Hz=Hz)
  bank.reset()
  x = np.random.default_rng(23).uniform(-1, 1, (S, N))
  y = bank.process(x, layout="chan")               # 8 MB in, 134 MB out
  k = alz.gammatone_erb_constants(4)[0]
  for band in (0, 7, 15):
    filt = alz.gammatone.slaney(fcs[band], k * alz.erb(fcs[band], Hz))
    secs = [(f.numlist, f.denlist) for f in filt]
    ref = oracle.bank([len(q[0]) for q in secs], [len(q[1]) for q in secs], np.concatenate([q[0] for q in secs]),
                      np.concatenate([q[1] for q in secs]), x, layout="chan")
    assert same_bits(y[band * S:(band + 1) * S], ref)
