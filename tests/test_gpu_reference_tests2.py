"""More of the reference's own tests for the path and its neighbours, restated with
audiolazy_amd: TestEnvelope / TestMAverage (audiolazy/tests/test_analysis.py:118-143),
TestChunks (test_io.py:125-160), TestWavStream (test_wav.py:66-175), the noise / constant
sources (test_synth.py:146-190) and the autocorrelation-method LPC cases (test_lpc.py)."""
import io
import itertools
import struct
from tempfile import NamedTemporaryFile

import pytest

from test_reference_algebra import almost_eq

pytestmark = pytest.mark.gpu
p = pytest.mark.parametrize
inf = float("inf")


@pytest.fixture(scope="module")
def al():
  import audiolazy_amd
  assert audiolazy_amd.device_count() >= 1
  return audiolazy_amd


# ---------------------------------------------------------------- test_analysis.py:118-143
@p("name", ["rms", "abs", "squared"])
def test_envelope_always_positive_and_keep_size(al, name):
  sig = [-5, -2.2, -1, 2, 4., 5., -1., -1.8, -22, -57., 1., 12.]
  out_stream = al.envelope[name](sig)
  assert isinstance(out_stream, al.Stream)
  out_list = list(out_stream)
  assert len(out_list) == len(sig) and all(el >= 0. for el in out_list)


@p("val", [0, 1, 2, 3., 4.8])
@p("size", [2, 8, 15, 23])
@p("name", ["recursive", "fir"])
def test_maverage_const_input(al, val, size, name):
  result = al.maverage[name](size)(al.Stream(val))
  small_result = result.take(size - 1)
  ramp = [val * i / size for i in range(size)][1:]              # list(line(size, 0., val))[1:]
  assert almost_eq(small_result, ramp)
  for el in result.take(int(2.5 * size)):
    assert abs(el - val) <= 2 ** -23 * abs(el + val)


# -------------------------------------------------------------------- test_io.py:125-160
chunk_data = [17., -3.42, 5.4, 8.9, 27., 45.2, 1e-5, -3.7e-4, 7.2, .8272, -4.]
ld = len(chunk_data)
chunk_sizes = [1, 2, 3, 4, ld - 1, ld, ld + 1, 2 * ld, 2 * ld + 1]


@p("name", ["struct", "array"])
@p("size", chunk_sizes)
@p("n_given", range(ld))
def test_chunks(al, name, size, n_given):
  given_data = chunk_data[:n_given]
  dfmt, padval = "f", 0.
  data = b"".join(al.chunks[name](given_data, size=size, dfmt=dfmt, padval=padval))
  samples_in = len(given_data)
  samples_out = samples_in
  if samples_in % size != 0:
    samples_out -= samples_in % -size
    assert samples_out > samples_in
  restored = struct.Struct(dfmt * samples_out).unpack(data)
  assert almost_eq(given_data, restored[:samples_in])
  assert almost_eq([padval] * (samples_out - samples_in), restored[samples_in:])


@p("name", ["struct", "array"])
@p("size", chunk_sizes)
def test_chunks_default_size(al, name, size):
  func, chunks = al.chunks[name], al.chunks
  dsize = chunks.size
  assert list(func(chunk_data)) == list(func(chunk_data, size=dsize))
  try:
    chunks.size = size
    assert list(func(chunk_data)) == list(func(chunk_data, size=size))
  finally:
    chunks.size = dsize


# -------------------------------------------------------------------- test_wav.py:38-175
def riff_chunk(chunk_id, *contents, **kwargs):
  payload = b"".join(contents)
  pad = b"\x00" if kwargs.get("align", True) and len(payload) % 2 else b""
  return chunk_id + struct.pack("<I", len(payload)) + payload + pad


def wave_data(data, channels=1, bits=16, rate=44100):
  """A RIFF/WAVE file image with a PCM fmt chunk and the given raw data bytes."""
  width = bits // 8
  fmt = riff_chunk(b"fmt ", struct.pack("<HHIIHH", 1, channels, rate, rate * channels * width,
                                          channels * width, bits))
  return riff_chunk(b"RIFF", b"WAVE", fmt, riff_chunk(b"data", data, align=False))


@pytest.fixture(params=["bytes_io", "temp_file_obj", "temp_file_name"])
def wave_file(request):
  if request.param == "bytes_io":
    yield lambda *args, **kwargs: io.BytesIO(wave_data(*args, **kwargs))
    return
  with NamedTemporaryFile(mode="rb+") as f:
    def fixture_func(*args, **kwargs):
      f.write(wave_data(*args, **kwargs))
      if request.param == "temp_file_obj":
        f.seek(0)
        return f
      f.flush()
      return f.name
    yield fixture_func


@p(("bits", "rate", "channels"), [(8, 44100, 1), (16, 3000, 3), (24, 8000, 2), (32, 12, 8)])
def test_wav_load_file_empty(al, bits, rate, channels, wave_file):
  wav_stream = al.WavStream(wave_file(b"", channels=channels, bits=bits, rate=rate))
  assert isinstance(wav_stream, al.Stream)
  assert (wav_stream.bits, wav_stream.channels, wav_stream.rate) == (bits, channels, rate)
  assert list(wav_stream) == []


wav_params = [
  (8, 8000, b"\x08\x7f\x18\xfa\xea\xce\x00", (-120, -1, -104, 122, 106, 78, -128)),
  (16, 48000, b"\x08\x91\xf3\x18\xfa\x82\xe4\x2a\xce\x00", (-0x6ef8, 0x18f3, -0x7d06, 0x2ae4, 0xce)),
  (24, 12345, b"\x63\x91\x36\x40\x10\xb0\xfa\xc6\xd0\x80\x78\xaf\x19\x82\xce",
   (0x369163, -0x4fefc0, -0x2f3906, -0x508780, -0x317de7)),
  (32, 87654, b"\x1f\x85\x6b\x3e\x7b\x14\xae\xbe\x89\xd2\xde\x3a\x6c\x09\x79\xba\x9a\x6d\x41\x19",
   (0x3e6b851f, -0x4151eb85, 0x3aded289, -0x4586f694, 0x19416d9a)),
]


@p(("bits", "rate", "data", "expected"), wav_params)
@p("keep", [True, False, None])
@p("channels", [1, 2])
def test_wav_load_file(al, bits, rate, data, expected, keep, wave_file, channels):
  if channels == 2:
    data, expected, rate = data * 2, expected * 2, rate // 3
  kwargs = {} if keep is None else dict(keep=keep)
  wav_stream = al.WavStream(wave_file(data, bits=bits, rate=rate, channels=channels), **kwargs)
  assert isinstance(wav_stream, al.Stream)
  assert (wav_stream.bits, wav_stream.channels, wav_stream.rate) == (bits, channels, rate)
  multiplier = 1 << (wav_stream.bits - 1)
  if keep:
    dtype = int
    if bits == 8:
      min_value, max_value = 0, 255
      result = list(wav_stream.copy() - 128)
    else:
      min_value, max_value = -multiplier - 1, multiplier
      result = list(wav_stream.copy())
  else:
    dtype, min_value, max_value = float, -1, 1 - 1 / multiplier
    result = list(wav_stream.copy() * multiplier)
  ws = al.thub(wav_stream, 3)
  assert all(isinstance(el, dtype) for el in ws)
  assert all(ws >= min_value)
  assert all(ws <= max_value)
  assert almost_eq(result, expected)


# ------------------------------------------------------------------ test_synth.py:146-190
@p(("name", "value"), [("ones", 1.0), ("zeros", 0.0), ("zeroes", 0.0)])
def test_ones_zeros(al, name, value):
  func = getattr(al, name)
  assert isinstance(func(), al.Stream) and func().take(25) == [value] * 25
  assert func(inf).take(30) == [value] * 30
  for dur in (-1, 0, .4, .5, 1, 2, 10):
    assert list(func(dur)) == [value] * max(al.rint(dur), 0)


def test_white_noise(al):
  assert all(-1 <= el <= 1 for el in al.white_noise().take(27))
  for high in (1, 0, -.042):
    assert all(-1 <= el <= high for el in al.white_noise(inf, high=high).take(32))
  for dur, low in itertools.product((-1, 0, .4, .5, 1, 2, 10), (0, .17)):
    got = list(al.white_noise(dur, low=low))
    assert len(got) == max(al.rint(dur), 0) and all(low <= el <= 1 for el in got)


# ------------------------------------------------------------------------- test_lpc.py
block_alternate = [1., 1. / 2., -1. / 8., 1. / 32., -1. / 128., 1. / 256., -1. / 512., 1. / 1024., -1. / 4096.,
                   1. / 8192.]


def filt_almost_eq(f, g):
  """almost_eq on two filters = on their (numdict, dendict) pairs (LinearFilter.__iter__)."""
  return all(sorted(a) == sorted(b) and all(abs(a[k] - b[k]) <= 2 ** -23 * abs(a[k] + b[k]) for k in a)
             for a, b in zip(f, g))


def test_lpc_block_info_kautocor(al):                            # test_lpc.py:59-72, 124-133
  z = al.z
  filt = al.lpc.kautocor(block_alternate, 3)
  assert filt_almost_eq(filt, 1 - 0.457681292332 * z ** -1 + 0.297451538058 * z ** -2 - 0.162014679229 * z ** -3)
  assert abs(filt.error - 1.03182436137) <= 2 ** -23 * abs(filt.error + 1.03182436137)


def test_lpc_docstring_kautocor(al):                             # test_lpc.py:218-224
  z = al.z
  filt = al.lpc.kautocor([-1, 0, 1, 0] * 4, 2)
  assert filt_almost_eq(filt, 1 + 0.875 * z ** -2)
  assert all(abs(a - b) <= 1e-7 for a, b in zip(filt.numerator, [1, 0., .875]))
  assert abs(filt.error - 1.875) <= 2 ** -23 * (filt.error + 1.875)


def test_levinson_durbin_one_five_three(al):                     # test_lpc.py:280-290
  z = al.z
  filt = al.levinson_durbin([1, 5, 3])
  assert filt_almost_eq(filt, 1 - 5. / 12. * z ** -1 - 11. / 12. * z ** -2)
  err = (1 - (11. / 12.) ** 2) * (1 - 5 ** 2)
  assert abs(filt.error - err) <= 2 ** -23 * abs(filt.error + err)


def test_lpc_kautocor_all_blocks_and_orders(al):                 # test_lpc.py:138-166 (kautocor vs the oracle)
  from oracle import oracle
  z = al.z
  small_block = [-1, 0, 1.2, -1, -2.7, 3, 7.1, 9, 12.3]
  big_block = list((1 - 2 * z ** -1)(range(150), zero=0))
  for blk in ([1, 5, 3], [1, 2, 3, 3, 2, 1], small_block, block_alternate, big_block):
    for order in (1, 2, 3, 7, 17, 18):
      filt = al.lpc.kautocor(blk, order)
      coefs, err = oracle.levinson_durbin(oracle.acorr(blk, order), order)
      assert filt.error >= 0.
      assert all(abs(a - b) <= 1e-9 * max(1., abs(b)) for a, b in zip(filt.numlist, coefs))
      assert abs(filt.error - err) <= 1e-9 * max(1., abs(err))


# ---------------------------------------------------------------- test_analysis.py:177-207
amdf_signal = [1.0, 2.0, 3.0, 2.0, 1.0, 2.0, 3.0, 2.0, 1.0]


@p(("lag", "size", "expected"), [
  (1, 1, [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]), (2, 1, [1.0, 2.0, 2.0, 0.0, 2.0, 0.0, 2.0, 0.0, 2.0]),
  (3, 1, [1.0, 2.0, 3.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]), (1, 2, [0.5, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]),
  (2, 2, [0.5, 1.5, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]), (3, 2, [0.5, 1.5, 2.5, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0]),
  (1, 4, [0.25, 0.5, 0.75, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]), (2, 4, [0.25, 0.75, 1.25, 1.25, 1.5, 1.0, 1.0, 1.0, 1.0]),
  (3, 4, [0.25, 0.75, 1.5, 1.75, 1.75, 1.5, 1.0, 1.0, 1.0])])
def test_amdf_input_output_mapping(al, lag, size, expected):
  filt = al.amdf(lag, size)
  assert callable(filt) and almost_eq(list(filt(amdf_signal)), expected)


@p("size", [1, 12])
def test_amdf_lag_zero(al, size):
  sig = list(al.white_noise(200))
  assert list(al.amdf(lag=0, size=size)(sig, zero=0)) == [0 for _ in sig]


@p("val", [0, 1, 2, 3., 4.8])
@p("size", [2, 8, 15, 23])
def test_maverage_deque_const_input(al, val, size):              # :131-143, the default strategy
  assert al.maverage.default is al.maverage.deque
  result = al.maverage(size)(al.Stream(val))
  assert almost_eq(result.take(size - 1), [val * i / size for i in range(size)][1:])
  for el in result.take(int(2.5 * size)):
    assert abs(el - val) <= 2 ** -23 * abs(el + val)


def test_control_stream_steers_a_time_varying_filter(al):
  """examples/formants.py in miniature: a ControlStream behind a resonator frequency; with a small
  block size the change is heard at the next block (lazy_stream.py:436-462 doctest first)."""
  cs = al.ControlStream(7)
  res = al.Stream(1, 3) + cs
  assert res.take(5) == [8, 10, 8, 10, 8]
  cs.value = 9
  assert res.take(5) == [12, 10, 12, 10, 12]
  old = al.block_size()
  try:
    al.block_size(8)
    gain = al.ControlStream(1.)
    out = (gain * al.z ** -1)(al.Stream(1.))
    assert out.take(8) == [0.] + [1.] * 7
    gain.value = 3.
    assert out.take(8) == [3.] * 8
  finally:
    al.block_size(old)
