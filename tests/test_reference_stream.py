"""The reference's own Stream / thub tests (audiolazy/tests/test_stream.py) restated with
audiolazy_amd: the Stream returned by every filter call has to behave like the reference's."""
import itertools as it
import math
import operator
from collections import deque

import pytest

from audiolazy_amd import Stream, thub
from audiolazy_amd.stream import StreamTeeHub
from test_reference_algebra import almost_eq

p = pytest.mark.parametrize
inf, nan, pi = float("inf"), float("nan"), math.pi


def test_no_args():                                              # :40-42
  with pytest.raises(TypeError):
    Stream()


@p("is_iter", [True, False])
@p("list_", [list(range(5)), [], [None, Stream, almost_eq, 2.4], [1, "a", 4]])
def test_from_list(list_, is_iter):                              # :44-47
  assert list(Stream(iter(list_) if is_iter else list_)) == list_


@p("tuple_input", [("strange", "tuple with list as fill value".split()), (1,), tuple(), ("abc", [12, 15], 8j)])
def test_from_tuple(tuple_input):                                # :49-53
  assert tuple(Stream(tuple_input)) == tuple_input


@p("value", [0, 1, 17, 200])
def test_lazy_range(value):                                      # :55-57
  assert list(Stream(range(value))) == list(range(value))


def test_class_docstring():                                      # :59-63
  x, y = Stream(it.count()), Stream(3)
  assert (2 * x + y).take(8) == [3, 5, 7, 9, 11, 13, 15, 17]


def test_mixed_inputs():                                         # :65-67
  with pytest.raises(TypeError):
    Stream([1, 2, 3], 5)


@p("d2_type", [list, tuple, Stream])
@p("d1_iter", [True, False])
@p("d2_iter", [True, False])
def test_multiple_inputs_iterable_iterator(d2_type, d1_iter, d2_iter):   # :69-80
  data1 = [1, "a", 4]
  data2 = d2_type([7.5, "abc", "abd", 9, it.chain])
  d2copy = data2.copy() if isinstance(data2, Stream) else data2
  data = [iter(data1) if d1_iter else data1]
  data += [iter(data2) if d2_iter else data2]
  assert list(Stream(*data)) == list(it.chain(data1, d2copy))


def test_non_iterable_inputs():                                  # :82-90
  for _, x in zip(range(15), Stream(25)):
    assert x == 25
  data = [2j + 3, 7.5, 75, type(2), Stream]
  for si, di in zip(Stream(*data), data * 4):
    assert si == di


def test_init_docstrings():                                      # :92-102
  assert list(Stream([1, 2, 3]) + Stream([8, 5])) == [9, 7]
  x = Stream(1, 2, 3) + Stream(8, 5)
  assert x.take(6) == [9, 7, 11, 6, 10, 8] and x.take(6) == [9, 7, 11, 6, 10, 8]
  assert x.take(3) == [9, 7, 11] and x.take(5) == [6, 10, 8, 9, 7]
  assert x.take(15) == [11, 6, 10, 8, 9, 7, 11, 6, 10, 8, 9, 7, 11, 6, 10]


def test_copy():                                                 # :104-124
  a, b = Stream([1, 2, 3]), Stream([8, 5])
  c, d = a.copy(), b.copy()
  assert type(a) == type(c) and type(b) == type(d)
  assert id(a) != id(c) and id(b) != id(d)
  assert iter(a) != iter(c) and iter(b) != iter(d)
  assert list(a) == [1, 2, 3] and list(c) == [1, 2, 3]
  assert b.take() == 8 and d.take() == 8 and b.take() == 5 and d.take() == 5
  with pytest.raises(StopIteration):
    b.take()
  with pytest.raises(StopIteration):
    d.take()


@p(("stream_size", "hop", "block_size"), [(48, 11, 15), (12, 13, 13), (42, 5, 22), (72, 14, 3), (7, 7, 7), (12, 8, 8),
                                          (12, 1, 5)])
def test_blocks(stream_size, hop, block_size):                   # :126-152
  data = Stream(range(stream_size))
  data_copy = data.copy()
  myblocks = data.blocks(size=block_size, hop=hop)
  myblocks_rev = Stream(reversed(list(data_copy))).blocks(size=block_size, hop=hop)
  for idx, (x, y) in enumerate(zip(myblocks, myblocks_rev)):
    assert len(x) == block_size and len(y) == block_size
    startx = idx * hop
    stopx = startx + block_size
    assert list(x) == [(k if k < stream_size else 0.) for k in range(startx, stopx)]
    starty, stopy = stream_size - 1 - startx, stream_size - 1 - stopx
    assert list(y) == [max(k, 0.) for k in range(starty, stopy, -1)]


def test_unary_operators_and_binary_pow_xor():                   # :154-165
  a = +Stream([1, 2, 3])
  b = -Stream([8, 5, 2, 17])
  c = Stream(True) ^ Stream([True, False, None])
  d = b ** a
  assert d.take(3) == [-8, 25, -8]
  with pytest.raises(StopIteration):
    d.take()
  assert c.take(2) == [False, True]
  with pytest.raises(TypeError):
    c.take()


def test_getattr_with_methods_and_equalness_operator():          # :167-177
  data = "trying again with strings...a bizarre iterable"
  a = Stream(data)
  b = a.copy()
  c = Stream("trying again ", range(5), "string", "." * 4)
  d = [True for _ in "trying again "] + [False for _ in range(5)] + [True for _ in "string"] + \
      [False, True, True, True]
  assert list(a == c) == d
  assert "".join(list(b.upper())) == data.upper()


def test_getattr_with_non_callable_attributes():                 # :179-187
  data = Stream(1 + 2j, 5 + 3j) * Stream(1j, 8, 1 - 1j)
  real, imag = Stream(-2, 40, 3, -3, 8, 8), Stream(1, 24, 1, 5, 16, -2)
  assert data.copy().real.take(6) == real.copy().take(6)
  assert data.copy().imag.take(6) == imag.copy().take(6)
  sum_data = data.copy().real + data.copy().imag
  assert sum_data.take(6) == (real + imag).take(6)


def test_no_boolean_no_next():                                   # :189-204
  with pytest.raises(TypeError):
    bool(Stream(range(2)))
  assert not hasattr(Stream(2), "next") and not hasattr(Stream(2), "__next__")


def test_truediv():                                              # :193-200
  input1, input2 = [1, 5, 7., 3.3], [9.2, 10, 11, 4.9]
  data = operator.truediv(Stream(input1), Stream(input2))
  assert isinstance(data, Stream)
  assert list(data) == [x / y for x, y in zip(input1, input2)]


def test_peek_take():                                            # :206-226
  data = Stream([1, 4, 3, 2])
  assert data.peek(3) == [1, 4, 3] and data.peek() == 1 and data.take() == 1
  assert data.peek() == 4 and data.peek(3) == [4, 3, 2] and data.peek() == 4 and data.take() == 4
  assert data.peek(3) == [3, 2] and data.peek(3, tuple) == (3, 2)
  assert data.peek(inf, tuple) == (3, 2) and data.take(inf, tuple) == (3, 2)
  assert data.peek(1) == [] and data.take(1) == [] and data.take(inf) == []
  assert Stream([1, 4, 3, 2]).take(inf) == [1, 4, 3, 2]
  with pytest.raises(StopIteration):
    data.peek()
  with pytest.raises(StopIteration):
    data.take()


def test_skip():                                                 # :228-256
  data = Stream(5, Stream, .2)
  assert data.skip(1).peek(4) == [Stream, .2, 5, Stream] and data.peek(4) == [Stream, .2, 5, Stream]
  assert data.skip(3).peek(4) == [Stream, .2, 5, Stream] and data.peek(4) == [Stream, .2, 5, Stream]
  assert data.skip(2).peek(4) == [5, Stream, .2, 5] and data.peek(4) == [5, Stream, .2, 5]
  data = Stream(range(25))
  data2 = data.copy()
  assert data.skip(4).peek(4) == [4, 5, 6, 7] and data2.peek(4) == [0, 1, 2, 3]
  assert data2.skip(30).peek(4) == []
  memory = {"last": 0}

  def tg():
    while True:
      memory["last"] += 1
      yield memory["last"]
  data = Stream(tg())
  assert data.take(3) == [1, 2, 3]
  data.skip(7)
  assert memory["last"] == 3                                    # lazy
  assert data.take() == 11 and memory["last"] == 11


def test_limit():                                                # :258-274
  r = lambda *a: list(range(*a))
  assert Stream(range(25)).limit(10).take(inf) == r(10)
  for n in (40, 25, 26):
    assert Stream(range(25)).limit(n).take(inf) == r(25)
  assert Stream(range(25)).limit(24).take(inf) == r(24)
  assert Stream(range(45)).skip(2).limit(13).take(inf) == r(2, 15)
  assert Stream(range(45)).limit(13).skip(3).take(inf) == r(3, 13)
  for noise in (-.3, 0, .1):
    assert Stream(0, 1, 2).limit(7 + noise).peek(10) == [0, 1, 2, 0, 1, 2, 0]
    data = Stream(-1, .2, it)
    assert data.skip(2).limit(9 + noise).peek(15) == [it, -1, .2] * 3


@p("noise", [-.3, 0., .1])
def test_take_peek_skip_with_float(noise):                       # :276-289
  data = [1.2, 7.7, 1e-3, 1e-17, 2e8, 27.1, 14.003, 1.0001, 7.3e5, 0.]
  ds = Stream(data)
  assert ds.limit(5 + noise).peek(10 - noise) == data[:5]
  assert ds.skip(1 + noise).limit(3 - noise).peek(10 + noise) == data[1:4]
  ds = Stream(data)
  assert ds.skip(2 + noise).peek(20 + noise) == data[2:]
  assert ds.skip(3 - noise).peek(20 - noise) == data[5:]
  assert ds.skip(4 + noise).peek(1 + noise) == [data[9]]
  ds = Stream(data)
  assert ds.skip(4 - noise).peek(2 - noise) == data[4:6]
  assert ds.skip(1 - noise).take(2 + noise) == data[5:7]
  assert ds.peek(inf) == data[7:] and ds.take(inf) == data[7:]


def test_take_peek_special_counts():                             # :291-321
  data = list(range(30))
  ds1 = Stream(data)
  ds1.take(3)
  assert ds1.peek(inf) == data[3:] and ds1.take(inf) == data[3:]
  ds2 = Stream(data)
  ds2.take(4)
  assert ds2.peek(inf * 2e-18) == data[4:] and ds2.take(inf * 1e-25) == data[4:]
  ds3 = Stream(data)
  ds3.take(1)
  assert ds3.peek(inf * 43) == data[1:] and ds3.take(inf * 200) == data[1:]
  for dur in [nan, nan * 23, nan * -5, nan * .3, nan * -.18, nan * 0]:
    assert Stream(29).take(dur) == [] and Stream([23]).peek(dur) == []
  for dur in [-inf, inf * -1e-16, -1, -2, -.18, 0, 0., -0.]:
    assert Stream(1, 2, 3).take(dur) == [] and Stream(-1, -23).peek(dur) == []
  for dur in [Stream, it, [], (2, 3), 3j]:
    with pytest.raises(TypeError):
      Stream([]).take(dur)
    with pytest.raises(TypeError):
      Stream([128]).peek(dur)


def test_take_peek_none():                                       # :323-338
  items = [Stream, it, [], (2, 3), 3j]
  data = Stream(items)
  for item in items:
    peeked, taken = data.peek(), data.take()
    assert (peeked is item and taken is item) or (peeked == item and taken == item)
  with pytest.raises(StopIteration):
    data.peek()
  with pytest.raises(StopIteration):
    data.take()


@p("constructor", [list, tuple, set, deque])
def test_take_peek_constructor(constructor):                     # :340-353
  ds = Stream([1, 2, 3] * 12)
  assert ds.peek(constructor=constructor) == 1 and ds.take(constructor=constructor) == 1
  assert ds.peek(constructor=constructor) == 2 and ds.take(constructor=constructor) == 2
  assert ds.peek(3, constructor=constructor) == constructor([3, 1, 2])
  assert ds.take(4, constructor=constructor) == constructor([3, 1, 2, 3])
  remain = constructor([1, 2, 3] * 10)
  assert ds.peek(inf, constructor=constructor) == remain and ds.take(inf, constructor=constructor) == remain
  assert ds.peek(3, constructor=constructor) == constructor()
  assert ds.take(5, constructor=constructor) == constructor()


def test_abs():                                                  # :355-369
  data = [-1, 5, 2 + 3j, 8, -.7]
  ds = abs(Stream(data))
  assert isinstance(ds, Stream)
  assert ds.take(len(data)) == [abs(el) for el in data] and ds.take(1) == []
  assert abs(Stream([])).peek(9) == []
  assert abs(Stream([5, -12, 14j, -2j, 0])).take(inf) == [5, 12, 14., 2., 0]
  data = abs(Stream([1.2, -1.57e-3, -(pi ** 2), -2j, 8 - 4j]))
  assert almost_eq(data.peek(inf), [1.2, 1.57e-3, pi ** 2, 2., 4 * 5 ** .5])
  assert data.take() == 1.2
  assert abs(data.take() - 1.57e-3) <= 2 ** -23 * (2 * 1.57e-3)


map_filter_data = [list(range(5)), list(range(9, 0, -2)), [7, 22, -5], [8., 3., 15.], list(range(20, 40, 3))]


@p("data", map_filter_data)
@p("func", [lambda x: x ** 2, lambda x: x // 2, lambda x: 18])
def test_map(data, func):                                        # :380-393
  expected = [func(x) for x in data]
  assert list(Stream(data).map(func)) == expected
  dt = thub(data, 2)
  assert isinstance(dt, StreamTeeHub)
  dt_data = dt.map(func)
  assert isinstance(dt_data, Stream) and dt_data.take(inf) == expected
  assert list(dt.map(func)) == expected
  with pytest.raises(IndexError):
    dt.map(func)


@p("data", map_filter_data)
@p("func", [lambda x: x > 0, lambda x: x % 2 == 0, lambda x: False])
def test_filter(data, func):                                     # :395-408
  expected = [x for x in data if func(x)]
  assert list(Stream(data).filter(func)) == expected
  dt = thub(data, 2)
  dt_data = dt.filter(func)
  assert isinstance(dt_data, Stream) and dt_data.take(inf) == expected
  assert list(dt.filter(func)) == expected
  with pytest.raises(IndexError):
    dt.filter(func)


def test_thub_take_peek():                                       # :495-508
  data = thub(Stream(1, 2, 3).limit(50), 2)
  assert data.peek() == 1 and data.peek(.2) == [] and data.peek(1) == [1]
  with pytest.raises(AttributeError):
    data.take()
  assert data.peek(22) == Stream(1, 2, 3).take(22)
  assert data.peek(42.2) == Stream(1, 2, 3).take(42)
  with pytest.raises(AttributeError):
    data.take(2)
  assert data.peek(57.8) == Stream(1, 2, 3).take(50)
  assert data.peek(inf) == Stream(1, 2, 3).take(50)


@p("noise", [-.3, 0, .1])
def test_thub_limit(noise):                                      # :510-536
  source = [.1, -.2, 18, it, Stream]
  length = len(source)
  data = thub(Stream(*source).limit(4 * length), 3)
  first_copy = data.limit(length + noise)
  assert isinstance(first_copy, Stream) and not isinstance(first_copy, StreamTeeHub)
  assert list(first_copy) == source
  assert data.peek(3 - noise) == source[:3]
  assert Stream(data).take(inf) == 4 * source
  third_copy = data.limit(5 * length + noise)
  assert isinstance(third_copy, Stream) and not isinstance(third_copy, StreamTeeHub)
  assert third_copy.take(inf) == 4 * source
  assert isinstance(data, StreamTeeHub)
  with pytest.raises(IndexError):
    data.limit(3)


@p("noise", [-.3, 0, .1])
def test_thub_skip_append(noise):                                # :538-562
  source = [9, 14, -7, noise]
  length = len(source)
  data = thub(Stream(*source).limit(7 * length), 3)
  first_copy = data.skip(length + 1)
  assert isinstance(first_copy, Stream) and not isinstance(first_copy, StreamTeeHub)
  assert first_copy is first_copy.append([8])
  assert list(first_copy) == source[1:] + 5 * source + [8]
  assert data.skip(1 + noise).peek(3 - noise) == source[1:4]
  assert data.append([1]).skip(length - noise).take(inf) == 6 * source + [1]
  assert isinstance(data, StreamTeeHub)
  with pytest.raises(IndexError):
    data.skip(1)
  with pytest.raises(IndexError):
    data.append(3)


@p("size", [4, 5, 6])
@p("hop", [None, 1, 5])
def test_thub_blocks(size, hop):                                 # :564-576
  copies = 8 - size
  source = Stream(7, 8, 9, -1, -1, -1, -1).take(40)
  data = thub(source, copies)
  expected = list(Stream(source).blocks(size=size, hop=hop).map(list))
  for _ in range(copies):
    blks = data.blocks(size=size, hop=hop).map(list)
    assert isinstance(blks, Stream) and not isinstance(blks, StreamTeeHub)
    assert blks.take(inf) == expected
  with pytest.raises(IndexError):
    data.blocks(size=size, hop=hop)


# ---------------------------------------------------------------- Streamix (lazy_stream.py:633-724)
def test_streamix_doctests():
  from audiolazy_amd import Streamix, sHz
  smix = Streamix(zero=0)
  smix.add(0, [-1, 1, 3, 2])
  smix.add(2, Stream([4, 4, 4]))
  smix.add(0, tuple([-3, -5, -7, -5, -7, -1]))
  assert list(smix) == [-1, 1, 4, 1, -3, -5, -7, -1]
  s, Hz = sHz(10)
  dur = int(2 * s)
  sdata = [1 + (-1 - 1) * i / (dur - 1) for i in range(dur)]        # list(line(2 * s, 1, -1, finish=True))
  smix = Streamix()
  smix.add(0.0 * s, sdata)
  smix.add(0.5 * s, sdata)
  smix.add(1.0 * s, sdata)
  result = [round(sm, 2) for sm in smix]
  assert len(result) == 35 and 0.5 * s == 5.0
  assert result[:7] == [1.0, 0.89, 0.79, 0.68, 0.58, 1.47, 1.26]
  assert result[10:17] == [0.42, 0.21, 0.0, -0.21, -0.42, 0.37, 0.05] and result[-1] == -1.0
  with pytest.raises(ValueError):
    Streamix().add(-1, [1])


def test_streamix_keep_and_gaps():
  from audiolazy_amd import Streamix
  smix = Streamix(zero=.5)
  smix.add(0, [1., 2.])
  smix.add(4, [10.])                       # two samples of silence in between
  assert list(smix) == [1.5, 2.5, .5, .5, 10.5]
  smix = Streamix(keep=True)
  smix.add(1, [3.])
  assert smix.take(4) == [0., 3., 0., 0.]
