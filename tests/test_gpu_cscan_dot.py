"""The fused time-parallel cascade (csrc/alz_scan.hip, round 5): the zero-state pass as dot products with the cascade's
impulse responses (k_cdot), the chunk-state recursion by DPP broadcasts (k_cscan_fix), and time-major blocks -- the
reference's vector-valued samples, ``gammatone`` banks of lazy_auditory.py:158-218 fed as in
tests/test_filters_extdep.py:49-89.  Opt-in mode, not bit-exact by construction: the bar is 1e-9 normalised against
the oracle (the contract is 1e-6)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def alz():
  import audiolazy_amd
  audiolazy_amd.load_library()
  assert audiolazy_amd.device_count() >= 1, "no HIP device: the engine has no CPU path"
  return audiolazy_amd


@pytest.fixture(scope="module")
def oracle():
  from oracle import oracle as o
  return o


def norm_err(got, ref, axis):
  den = np.abs(ref).max(axis=axis)
  den[den == 0] = 1.0
  return float((np.abs(got - ref).max(axis=axis) / den).max())


def band_tables(alz, fcs, Hz, strategy, S):
  """(nbs, nas, b [bands * S, sum nb], a [bands * S, sum na]) of the bank, channel = band * S + stream."""
  k = alz.gammatone_erb_constants(4)[0]
  bands = [getattr(alz.gammatone, strategy)(fc, k * alz.erb(fc, Hz)) for fc in fcs]
  nbs = [max(len(band[s].numlist) for band in bands) for s in range(4)]
  nas = [max(len(band[s].denlist) for band in bands) for s in range(4)]

  def row(band, attr, sizes):
    out = []
    for s, n in enumerate(sizes):
      lst = list(getattr(band[s], attr))
      out += lst + [0.0] * (n - len(lst))
    return out
  b = np.repeat(np.array([row(band, "numlist", nbs) for band in bands]), S, axis=0)
  a = np.repeat(np.array([row(band, "denlist", nas) for band in bands]), S, axis=0)
  return nbs, nas, b, a


def make_bank(alz, bands, streams, strategy="slaney", fs=48000, lo=50., hi=20000.):
  s_, Hz = alz.sHz(fs)
  fcs = [f * Hz for f in alz.erb_space(lo, hi, bands)]
  bank = alz.gammatone_bank(fcs, streams, strategy=strategy, Hz=Hz)
  return bank, fcs, Hz


@pytest.mark.parametrize("strategy,streams,bands,n", [("slaney", 1, 256, 1 << 16), ("klapuri", 1, 64, 1 << 16),
                                                         ("slaney", 3, 128, 3 << 14), ("slaney", 2, 8, 1 << 15),
                                                         ("slaney", 1, 256, 272 * 256)])   # 17 tiles per chunk: an odd tile behind the stored pairs, segments that are not whole lines (k_cdot)
def test_dot_product_zero_state_pass(alz, oracle, strategy, streams, bands, n):
  """After reset the bank's state is self-consistent: the zero-state pass is k_cdot; block after block (the fix kernel
  leaves the bank's state itself), then a serial continuation from that state."""
  import torch
  bank, fcs, Hz = make_bank(alz, bands, streams, strategy)
  bank.set_time_parallel(True)
  bank.reset()
  nbs, nas, b, a = band_tables(alz, fcs, Hz, strategy, streams)
  rng = np.random.default_rng(11 * bands + streams)
  blocks = [rng.uniform(-1, 1, (streams, m)) for m in (n, 1 << 14, 1 << 15)]
  tail = rng.uniform(-1, 1, (streams, 777))
  ref = oracle.bank(nbs, nas, b, a, np.tile(np.concatenate(blocks + [tail], axis=1), (bands, 1)), layout="chan")
  at = 0
  for x in blocks:
    y = bank.process(torch.from_numpy(x).cuda(), layout="chan").cpu().numpy()
    assert "k_cscan(k_cdot" in bank.last_kernel, bank.last_kernel
    assert norm_err(y, ref[:, at:at + x.shape[1]], 1) <= 1e-9
    at += x.shape[1]
  bank.set_time_parallel(False)
  y = bank.process(torch.from_numpy(tail).cuda(), layout="chan").cpu().numpy()
  assert norm_err(y, ref[:, at:], 1) <= 1e-9


def test_set_state_consistency_decides_the_pass(alz, oracle):
  """An arbitrary set_state (section s + 1's input history differs from section s's output history) takes the cascade
  kernel for the zero-state pass -- chunk 0 then starts from the state as given; the block after it is consistent
  again.  A set_state whose rows ARE consistent keeps the dot-product pass.  Both against the oracle from that state."""
  import torch
  bands, streams, n = 64, 1, 1 << 15
  bank, fcs, Hz = make_bank(alz, bands, streams)
  bank.set_time_parallel(True)
  nbs, nas, b, a = band_tables(alz, fcs, Hz, "slaney", streams)
  C = bands * streams
  rng = np.random.default_rng(5)
  thx, thy = sum(nb - 1 for nb in nbs), sum(na - 1 for na in nas)
  for consistent in (False, True):
    xh, yh = rng.uniform(-1e-3, 1e-3, (C, thx)), rng.uniform(-1e-3, 1e-3, (C, thy))
    if consistent:                      # x history of section s + 1 := y history of section s
      ox, oy = nbs[0] - 1, 0
      for s in range(1, 4):
        xh[:, ox:ox + nbs[s] - 1] = yh[:, oy:oy + nbs[s] - 1]
        ox += nbs[s] - 1
        oy += nas[s - 1] - 1
    bank.reset()
    bank.set_state(xh, yh)
    x1, x2 = rng.uniform(-1, 1, (streams, n)), rng.uniform(-1, 1, (streams, n))
    rxh, ryh = xh.copy(), yh.copy()
    ref1 = oracle.bank(nbs, nas, b, a, np.tile(x1, (bands, 1)), layout="chan", xh=rxh, yh=ryh)
    ref2 = oracle.bank(nbs, nas, b, a, np.tile(x2, (bands, 1)), layout="chan", xh=rxh, yh=ryh)
    y1 = bank.process(torch.from_numpy(x1).cuda(), layout="chan").cpu().numpy()
    assert ("k_cdot" in bank.last_kernel) == consistent and "k_cscan" in bank.last_kernel, bank.last_kernel
    assert norm_err(y1, ref1, 1) <= 1e-9
    y2 = bank.process(torch.from_numpy(x2).cuda(), layout="chan").cpu().numpy()
    assert "k_cscan(k_cdot" in bank.last_kernel, bank.last_kernel
    assert norm_err(y2, ref2, 1) <= 1e-9
    # the state on the device after two blocks, against the oracle's (same bar: it went through the chunk recursion)
    gxh, gyh = bank.get_state()
    assert np.max(np.abs(gyh - ryh)) <= 1e-9 * max(1.0, np.max(np.abs(ryh)))
    assert np.max(np.abs(gxh - rxh)) <= 1e-9 * max(1.0, np.max(np.abs(rxh)))


@pytest.mark.parametrize("strategy,streams,bands,n,dot", [("slaney", 1, 256, 1 << 16, True), ("klapuri", 1, 64, 1 << 16, True),
                                                             ("slaney", 64, 16, 1 << 14, False), ("slaney", 128, 4, 1 << 14, False)])
def test_time_major_blocks(alz, oracle, strategy, streams, bands, n, dot):
  """Rows = samples (the reference's vector-valued items): one stream through every band -- [N, 1] in, [N, bands] out,
  the input broadcast inside the cascade kernel -- and banks whose streams come in groups of 64."""
  import torch
  bank, fcs, Hz = make_bank(alz, bands, streams, strategy)
  bank.set_time_parallel(True)
  bank.reset()
  nbs, nas, b, a = band_tables(alz, fcs, Hz, strategy, streams)
  rng = np.random.default_rng(3 * bands + streams)
  x1, x2 = rng.uniform(-1, 1, (n, streams)), rng.uniform(-1, 1, (n, streams))
  xt = np.concatenate([x1, x2], axis=0)
  ref = oracle.bank(nbs, nas, b, a, np.tile(np.ascontiguousarray(xt.T), (bands, 1)), layout="chan").T   # [N, bands * streams]
  y1 = bank.process(torch.from_numpy(x1).cuda(), layout="time").cpu().numpy()
  assert "k_cscan" in bank.last_kernel and ("k_cdot" in bank.last_kernel) == dot, bank.last_kernel
  assert y1.shape == (n, bands * streams)
  assert norm_err(y1, ref[:n], 0) <= 1e-9
  y2 = bank.process(torch.from_numpy(x2).cuda(), layout="time").cpu().numpy()
  assert "k_cscan" in bank.last_kernel, bank.last_kernel
  assert norm_err(y2, ref[n:], 0) <= 1e-9


def test_one_stream_channel_major_takes_the_broadcast_replay(alz, oracle):
  """[1, N] -> [bands, N]: when the chunks fill the chip (>= 1024 groups of 64 virtual channels) the virtual channels are
  chunk-major in this layout too -- 64 adjacent BANDS of one chunk per wave, the one input row read by wave-uniform
  loads (k_casc<bc>: no input tiles travel); with fewer groups the chunks of a channel stay the lanes (k_pipe)."""
  import torch
  bank, fcs, Hz = make_bank(alz, 256, 1)
  nbs, nas, b, a = band_tables(alz, fcs, Hz, "slaney", 1)
  rng = np.random.default_rng(77)
  x = rng.uniform(-1, 1, (1, 3 << 16))
  ref = oracle.bank(nbs, nas, b, a, np.tile(x, (256, 1)), layout="chan")
  for tp, n0, n1, want in ((True, 0, 1 << 16, "k_cscan(k_cdot+k_casc<bc>)"), (True, 1 << 16, 2 << 16, "k_cscan(k_cdot+k_casc<bc>)"),
                           (1024, 2 << 16, 3 << 16, "k_cscan(k_cdot+k_pipe)")):
    bank.set_time_parallel(tp)
    if n0 == 0:
      bank.reset()
    y = bank.process(torch.from_numpy(np.ascontiguousarray(x[:, n0:n1])).cuda(), layout="chan").cpu().numpy()
    assert bank.last_kernel == want, bank.last_kernel
    assert norm_err(y, ref[:, n0:n1], 1) <= 1e-9


def test_time_major_matches_channel_major(alz):
  """The same one-stream block in both layouts: the chunk states come out of the same kernels (k_cdot, k_cscan_fix) and
  the replay of every chunk is the same arithmetic, so the two results are the same doubles."""
  import torch
  bank, fcs, Hz = make_bank(alz, 128, 1)
  bank.set_time_parallel(4096)
  x = np.random.default_rng(8).uniform(-1, 1, (1 << 18,))
  bank.reset()
  yc = bank.process(torch.from_numpy(x[None, :].copy()).cuda(), layout="chan").cpu().numpy()
  kc = bank.last_kernel
  bank.reset()
  yt = bank.process(torch.from_numpy(x[:, None].copy()).cuda(), layout="time").cpu().numpy()
  assert "k_cdot" in kc and "k_cdot" in bank.last_kernel, (kc, bank.last_kernel)
  assert np.array_equal(yc.T.view(np.uint64), yt.view(np.uint64))
