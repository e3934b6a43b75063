"""CPU side of the ingest / egress formats and the ParallelFilter sum: the oracle restatements
against reference-generated vectors (tests/golden/formats.json, made by oracle/gen_golden.py
from WavStream lazy_wav.py:31-130, chunks.struct lazy_io.py:48-94 and ParallelFilter.__call__
lazy_filters.py:1048-1054), and the host logic that needs no GPU."""
import array
import struct

import numpy as np
import pytest

from conftest import load_golden, unhex
from oracle import oracle


def same_bits(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return a.shape == b.shape and bool(np.all(a.view(np.uint64) == b.view(np.uint64)))


@pytest.mark.parametrize("idx", range(8))
def test_oracle_pcm_decode_matches_wavstream(idx):
  c = load_golden("formats.json")["wav"][idx]
  raw = bytes.fromhex(c["raw"])
  assert same_bits(oracle.pcm_decode(raw, c["bits"]), unhex(c["scaled"]))
  assert oracle.pcm_decode(raw, c["bits"], keep=True).tolist() == [float(v) for v in c["kept"]]


def test_oracle_pcm_encode_matches_chunks_struct():
  for c in load_golden("formats.json")["chunks"]:
    x = c["x"] if c["ints"] else unhex(c["x"])
    pad = c["padval"] if c["ints"] else float.fromhex(c["padval"])
    full = list(x) + [pad] * ((-len(x)) % c["size"])
    assert oracle.pcm_encode(full, c["dfmt"], c["byte_order"]).hex() == "".join(c["chunks"])


def test_oracle_pcm_encode_is_struct_pack():
  rng = np.random.default_rng(3)
  x = np.concatenate([rng.uniform(-1, 1, 500), [0.0, -0.0, 1e-46, 3.4028234e38, np.inf, -np.inf]])
  for order in ("<", ">"):
    assert oracle.pcm_encode(x, "f", order) == struct.pack("%s%df" % (order, x.size), *x.tolist())
    assert oracle.pcm_encode(x, "d", order) == struct.pack("%s%dd" % (order, x.size), *x.tolist())
  xi = rng.integers(-32768, 32768, 300)
  assert oracle.pcm_encode(xi, "h", "<") == struct.pack("<300h", *xi.tolist())
  assert oracle.pcm_encode(xi, "h", None) == array.array("h", xi.tolist()).tobytes()


@pytest.mark.parametrize("idx", range(4))
def test_oracle_parallel_sum_matches_reference(idx):
  c = load_golden("formats.json")["parallel"][idx]
  x = unhex(c["x"])
  memory = None if c["memory"] is None else unhex(c["memory"])
  zero = float.fromhex(c["zero"])
  outs = [oracle.df1(unhex(s["b"]), unhex(s["a"]), x, memory=memory, zero=zero) for s in c["sections"]]
  y = oracle.mix(np.stack(outs, axis=1), len(outs), 1, layout="time")[:, 0]
  assert same_bits(y, unhex(c["y"]))
  yc = oracle.mix(np.stack(outs, axis=0), len(outs), 1, layout="chan")[0]
  assert same_bits(yc, unhex(c["y"]))


def test_chunks_host_logic_without_gpu():
  from audiolazy_amd import pcm
  assert pcm.chunks.size == 2048                      # lazy_io.py:45
  assert pcm.chunks.default is pcm.chunks.struct
  with pytest.raises(struct.error):
    pcm._big_endian("?")
  assert pcm._big_endian(None) == 0 and pcm._big_endian(">") == 1 and pcm._big_endian("!") == 1
  # integer formats refuse floats before anything reaches the device (struct.pack does the same)
  with pytest.raises(struct.error):
    list(pcm.chunks.struct([1.0, 2.0], size=2, dfmt="h"))
  with pytest.raises(TypeError):
    list(pcm.chunks.array([1.0, 2.0], size=2, dfmt="h"))
  with pytest.raises(NotImplementedError):
    list(pcm.chunks([1, 2], size=2, dfmt="q"))
