"""WavStream / chunks / decode_pcm / encode_pcm and the ParallelFilter mixdown on the GPU,
through the C ABI (alz_pcm_decode_dev, alz_pcm_encode_dev, alz_mix_dev).  Bit-exact against
reference-generated vectors (tests/golden/formats.json) and, at larger sizes, the oracle."""
import array
import io
import struct
import wave

import numpy as np
import pytest

from conftest import load_golden, unhex
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def al():
  import audiolazy_amd
  assert audiolazy_amd.device_count() >= 1
  return audiolazy_amd


@pytest.fixture(scope="module")
def pcm(al):
  from audiolazy_amd import pcm as module
  return module


def same_bits(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return a.shape == b.shape and bool(np.all(a.view(np.uint64) == b.view(np.uint64)))


@pytest.mark.parametrize("idx", range(8))
def test_wavstream_matches_reference(pcm, idx, tmp_path):
  c = load_golden("formats.json")["wav"][idx]
  path = tmp_path / "in.wav"
  path.write_bytes(bytes.fromhex(c["file"]))
  ws = pcm.WavStream(str(path))
  assert (ws.rate, ws.channels, ws.bits) == (c["rate"], c["channels"], c["bits"])
  got = list(ws)
  assert all(isinstance(v, float) for v in got)
  assert same_bits(got, unhex(c["scaled"]))
  kept = list(pcm.WavStream(str(path), keep=True, block_frames=5))   # several device blocks
  assert all(isinstance(v, int) for v in kept) and kept == c["kept"]
  # file-like objects work too (wave.open accepts them), and blocks(channels) regroups frames
  ws = pcm.WavStream(io.BytesIO(bytes.fromhex(c["file"])))
  frames = [list(b) for b in ws.blocks(ws.channels)]
  assert same_bits(np.array(frames).ravel(), unhex(c["scaled"]))
  arr = pcm.WavStream(str(path)).array()
  assert arr.shape == (len(c["scaled"]) // c["channels"], c["channels"])
  assert same_bits(arr.ravel(), unhex(c["scaled"]))


@pytest.mark.parametrize("bits", [8, 16, 24, 32])
@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 1023, 100003])
def test_decode_pcm_vs_oracle(pcm, bits, n):
  rng = np.random.default_rng(bits * 1000 + n)
  raw = rng.integers(0, 256, n * bits // 8, dtype=np.uint8).tobytes()
  for keep in (False, True):
    assert same_bits(pcm.decode_pcm(raw, bits, keep=keep), oracle.pcm_decode(raw, bits, keep=keep))


def test_decode_pcm_torch_and_into_the_bank(al, pcm):
  import torch
  rng = np.random.default_rng(9)
  C, N = 64, 8192
  raw = rng.integers(0, 256, N * C * 2, dtype=np.uint8)
  x = pcm.decode_pcm(torch.from_numpy(raw).cuda(), 16).reshape(N, C)     # interleaved == time-major
  ref_x = oracle.pcm_decode(raw.tobytes(), 16).reshape(N, C)
  assert same_bits(x.cpu().numpy(), ref_x)
  s, Hz = al.sHz(48000)
  filt = al.resonator.z_exp(1000 * Hz, 100 * Hz)
  bank = al.FilterBank([(filt.numlist, filt.denlist)], n_inputs=C)
  bank.reset()
  y = bank.process(x.contiguous(), layout="time").cpu().numpy()
  ref = oracle.bank([3], [3], np.array(filt.numlist), np.array(filt.denlist), ref_x, layout="time")
  assert same_bits(y, ref)


def test_chunks_match_reference(pcm):
  for c in load_golden("formats.json")["chunks"]:
    x = c["x"] if c["ints"] else unhex(c["x"])
    pad = c["padval"] if c["ints"] else float.fromhex(c["padval"])
    for look in (1, 3):
      got = list(pcm.chunks.struct(iter(x), size=c["size"], dfmt=c["dfmt"], byte_order=c["byte_order"],
                                   padval=pad, lookahead=look))
      assert [g.hex() for g in got] == c["chunks"], (c["dfmt"], c["byte_order"], look)
    if c["byte_order"] is None:   # the array strategy is native-order only (lazy_io.py:97-128)
      got = list(pcm.chunks.array(iter(x), size=c["size"], dfmt=c["dfmt"], padval=pad))
      assert [g.hex() for g in got] == c["chunks"]
  assert len(next(iter(pcm.chunks([0.5] * 10)))) == 2048 * 4       # default size / format


@pytest.mark.parametrize("dfmt", ["b", "B", "h", "H", "i", "I", "f", "d"])
@pytest.mark.parametrize("order", [None, "<", ">"])
def test_encode_pcm_vs_oracle(pcm, dfmt, order):
  rng = np.random.default_rng(ord(dfmt))
  for n in (1, 4, 7, 4099):
    if dfmt in "fd":
      x = rng.uniform(-1, 1, n) * 10.0 ** rng.integers(-40, 38, n)
    else:
      lo, hi = pcm._INT_RANGE[dfmt]
      x = rng.integers(lo, hi + 1, n).astype(np.float64)
      x[0] = lo
      x[-1] = hi
    assert pcm.encode_pcm(x, dfmt, order) == oracle.pcm_encode(x, dfmt, order), (dfmt, order, n)
  if dfmt in "fd":
    special = np.array([0.0, -0.0, np.inf, -np.inf, 1e-46, -1e-46, 2.0 ** -149, 2.0 ** -150, 3.4028234e38])
    want = struct.pack("%s%d%s" % (order or "<", special.size, dfmt), *special.tolist())
    assert pcm.encode_pcm(special, dfmt, order) == want


def test_encode_pcm_errors_like_struct(pcm):
  with pytest.raises(OverflowError):                       # struct.pack("f", 1e39)
    pcm.encode_pcm(np.array([0.0, 1e39, 0.0, 0.0, 0.0]), "f")
  with pytest.raises(struct.error):                        # struct.pack("h", 40000)
    pcm.encode_pcm(np.array([1.0, 40000.0]), "h")
  with pytest.raises(struct.error):
    pcm.encode_pcm(np.array([1.0, 2.0, 3.0, -1.0]), "B")
  with pytest.raises(struct.error):                        # not an integer
    pcm.encode_pcm(np.array([1.0, 2.5, 3.0, 4.0, 5.0]), "i")
  with pytest.raises(struct.error):
    list(pcm.chunks.struct([1, 2, 70000], size=4, dfmt="h", padval=0))
  with pytest.raises(OverflowError):                       # array("h", [70000])
    list(pcm.chunks.array([1, 2, 70000], size=4, dfmt="h", padval=0))
  with pytest.raises(TypeError):                           # the float padval reaches array("h")
    list(pcm.chunks.array([1, 2, 3], size=4, dfmt="h"))
  # the array module stores inf for a too large "f" item instead of raising
  got = list(pcm.chunks.array([1e39, 1.0], size=2, dfmt="f"))
  assert got == [array.array("f", [float("inf"), 1.0]).tobytes()]


def test_encode_pcm_torch(pcm):
  import torch
  x = torch.linspace(-1, 1, 1001, dtype=torch.float64, device="cuda")
  got = pcm.encode_pcm(x, "f").cpu().numpy().tobytes()
  assert got == oracle.pcm_encode(x.cpu().numpy(), "f")


def test_wav_roundtrip_through_chunks(pcm, tmp_path):
  """WavStream(keep) -> chunks("h") reproduces the file's data bytes."""
  rng = np.random.default_rng(5)
  data = rng.integers(-32768, 32768, 5000).astype("<i2").tobytes()
  path = str(tmp_path / "rt.wav")
  w = wave.open(path, "wb")
  w.setnchannels(2); w.setsampwidth(2); w.setframerate(8000); w.writeframes(data); w.close()
  out = b"".join(pcm.chunks(pcm.WavStream(path, keep=True), size=1000, dfmt="h", byte_order="<", lookahead=2))
  assert out == data


# ---------------------------------------------------------------------------------- mixdown
@pytest.mark.parametrize("idx", range(4))
def test_parallel_filter_matches_reference(al, idx):
  c = load_golden("formats.json")["parallel"][idx]
  branches = [al.ZFilter(unhex(s["b"]), unhex(s["a"])) for s in c["sections"]]
  kw = dict(zero=float.fromhex(c["zero"]))
  if c["memory"] is not None:
    kw["memory"] = unhex(c["memory"])
  got = list(al.ParallelFilter(branches)(unhex(c["x"]), block=64, **kw))
  assert same_bits(got, unhex(c["y"]))


def test_parallel_filter_edge_cases(al):
  z = al.z
  assert list(al.ParallelFilter()([1, 2, 3], zero=0.5)) == [0.5, 0.5, 0.5]      # lazy_filters.py:1049-1051
  par = al.ParallelFilter(z ** -1, 1 - z ** -2)                                 # doctest :1040-1045 shape
  assert list(par([1., 2., 3., 4.])) == [1., 3., 4., 5.]
  with pytest.raises(ValueError):
    al.ParallelFilter(z, 1 + z ** -1)([1., 2.])
  with pytest.raises(ValueError):       # 1 / z ** -1 is stored as z: non-causal (lazy_filters.py:126-132)
    al.ParallelFilter(al.ZFilter([1.], [0., 1.]), 1 + z ** -1)([1., 2.])


@pytest.mark.parametrize("layout", ["time", "chan"])
@pytest.mark.parametrize("n_sets,n_inputs,n", [(1, 5, 33), (3, 1, 1000), (9, 7, 257), (20, 64, 512), (256, 4, 128)])
def test_mixdown_vs_oracle(al, layout, n_sets, n_inputs, n):
  import torch
  from audiolazy_amd.bank import mix_sets
  rng = np.random.default_rng(n_sets * 100 + n_inputs)
  shape = (n, n_sets * n_inputs) if layout == "time" else (n_sets * n_inputs, n)
  y = rng.uniform(-1, 1, shape) * 10.0 ** rng.integers(-8, 8, shape)
  ref = oracle.mix(y, n_sets, n_inputs, layout=layout)
  assert same_bits(mix_sets(y, n_sets, n_inputs, layout=layout), ref)
  got = mix_sets(torch.from_numpy(y).cuda(), n_sets, n_inputs, layout=layout)
  assert same_bits(got.cpu().numpy(), ref)


def test_gammatone_bank_mixdown(al):
  """Sum over the bands of an OUTER bank (the resynthesis-style mix of a filterbank)."""
  s, Hz = al.sHz(48000)
  B, S, N = 12, 5, 2048
  fcs = [f * Hz for f in al.erb_space(100., 8000., B)]
  bank = al.gammatone_bank(fcs, S, strategy="slaney", Hz=Hz)
  bank.reset()
  x = np.random.default_rng(1).uniform(-1, 1, (S, N))
  y = bank.process(x, layout="chan")
  assert same_bits(bank.mixdown(y, layout="chan"), oracle.mix(y, B, S, layout="chan"))
  with pytest.raises(ValueError):
    al.FilterBank([([1.], [1.])], n_inputs=3).mixdown(np.zeros((4, 3)))


# ------------------------------------------------------------------------------- Streamix
def test_mix_tracks_is_streamix(al):
  """alz_mix_tracks_dev against the host Streamix (same adds, same order): bit for bit, with float
  deltas, gaps, empty tracks and more tracks than one launch takes."""
  import torch
  rng = np.random.default_rng(11)
  for n_tracks, zero in ((0, 0.), (1, 0.), (3, .25), (7, -1.5), (60, 0.)):
    tracks = [rng.uniform(-1, 1, int(rng.integers(0, 400))) * 10.0 ** rng.integers(-6, 6) for _ in range(n_tracks)]
    deltas = [float(rng.choice([0, 0., .4, 1, 2.5, 17, 100.25, 300])) for _ in range(n_tracks)]
    smix = al.Streamix(zero=zero)
    for d, t in zip(deltas, tracks):
      smix.add(d, t.tolist())
    ref = list(smix)
    got = al.mix_tracks(tracks, deltas, zero=zero)
    assert same_bits(got, ref), (n_tracks, zero)
    if n_tracks:
      got = al.mix_tracks([torch.from_numpy(t).cuda() for t in tracks], deltas, zero=zero)
      assert same_bits(got.cpu().numpy(), ref)
  with pytest.raises(ValueError):
    al.mix_tracks([np.zeros(3)], [-1])
