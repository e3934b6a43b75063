"""The reference's own TestZFilter cases that *call* filters (audiolazy/tests/test_filters.py
:40-330), restated with audiolazy_amd in place of audiolazy: same data, same expressions, same
expectations and the same tolerance helper.  Every call below runs on the GPU engine."""
import itertools
import operator

import pytest

from test_reference_algebra import almost_eq, alpha

pytestmark = pytest.mark.gpu
p = pytest.mark.parametrize

data = [-7, 3] + list(range(10)) + [-50, 0] + list(range(70, -70, -11))   # only ints (:41)


@pytest.fixture(scope="module")
def al():
  import audiolazy_amd
  assert audiolazy_amd.device_count() >= 1
  return audiolazy_amd


def test_z_identity(al):                                         # :45-47
  assert list((al.z ** 0)(data)) == data


@p("amp", [-10, 3, 0, .2, 8])
def test_z_simple_amplification(al, amp):                        # :49-56
  z = al.z
  expected = [amp * di for di in data]
  op = operator.eq if isinstance(amp, int) else almost_eq
  assert op(list((amp * z ** 0)(data)), expected)
  assert op(list((z ** 0 * amp)(data)), expected)


@p("delay", range(1, 5))
def test_z_int_delay(al, delay):                                 # :59-62
  assert list((+al.z ** -delay)(data)) == [0] * delay + data[:-delay]


@p("amp", [1, -105, 43, 0, .128, 18])
@p("delay", range(1, 7))
def test_z_int_delay_with_amplification(al, amp, delay):         # :64-72
  z = al.z
  expected = [amp * di for di in ([0.] * delay + data[:-delay])]
  op = operator.eq if isinstance(amp, int) else almost_eq
  assert op(list((amp * z ** -delay)(data)), expected)
  assert op(list((z ** -delay * amp)(data)), expected)


def test_z_fir_size_2(al):                                       # :74-77
  expected = [a + b for a, b in zip(data, [0] + data[:-1])]
  assert list((1 + al.z ** -1)(data)) == expected


def test_z_fir_size_2_hybrid_amplification(al):                  # :79-83
  expected = [6. * a - 10 * b for a, b in zip(data, [0.] + data[:-1])]
  assert almost_eq((2 * (3. - 5 * al.z ** -1))(data), expected)


@p("num_delays", range(1, 5))
def test_z_many_fir_sizes_and_amplifications(al, num_delays):    # :85-99 (one amplitude tuple per size)
  z = al.z
  amps = [1, -15, 45, 0, .81, 17]
  for amp in itertools.islice(itertools.combinations_with_replacement(amps, num_delays + 1), 0, None, 7):
    filt = sum(amp[d] * z ** -d for d in range(num_delays + 1))
    parts = [[amp[d] * v for v in (z ** -d)(data)] for d in range(num_delays + 1)]
    expected = [sum(vals) for vals in zip(*parts)]
    assert almost_eq(filt(data), expected)


def test_z_fir_multiplication(al):                               # :101-105
  z = al.z
  filt = 8 * (2 * z ** -3 - 5 * z ** -4) * z ** 2 * 7
  expected = [56 * 2 * a - 56 * 5 * b for a, b in zip([0] + data[:-1], [0, 0] + data[:-2])]
  assert list(filt(data)) == expected


@p("a", alpha)
def test_z_one_pole(al, a):                                      # :107-113
  expected = [x for x in data]
  for i in range(1, len(expected)):
    expected[i] -= a * expected[i - 1]
  assert almost_eq((1 / (1 + a * al.z ** -1))(data), expected)


def test_z_power_alone(al):                                      # :135-140
  z = al.z
  for filt in (1 / z / 1, (1 / z) ** 1, (1 / z ** -1) ** -1):
    assert almost_eq(filt(data), [0.] + data[:-1])


@p("a", [a for a in alpha if a != 0])
@p("zero", [0., 0])
def test_z_truediv_unit_delay_divided_by_constant(al, a, zero):  # :142-151
  for el in [a, int(10 * a)]:
    expected = [x / a for x in [zero] + data[:-1]]
    assert almost_eq((al.z ** -1 / a)(data, zero=zero), expected)


@p("a", alpha)
def test_z_truediv_constant_over_delay(al, a):                   # :153-162
  expected = [a * x for x in data]
  for i in range(1, len(expected)):
    expected[i] -= expected[i - 1]
  assert almost_eq((a / (1 + al.z ** -1))(data), expected)


def test_z_power_with_denominator(al):                           # :164-172  y[n] = x[n-1] - y[n-2]
  z = al.z
  filt = (z ** -1 / (1 + z ** -2)) ** 1
  expected, mem2, mem1, xlast = [], 0., 0., 0.
  for di in data:
    newy = xlast - mem2
    mem2, mem1, xlast = mem1, newy, di
    expected.append(newy)
  assert almost_eq(filt(data), expected)


@p("a", alpha)
def test_z_one_pole_neg_afterwards(al, a):                       # :201-209
  expected = [x for x in data]
  for i in range(1, len(expected)):
    expected[i] -= a * expected[i - 1]
  assert almost_eq((-(1 / (1 + a * al.z ** -1)))(data), [-x for x in expected])


@p("a", alpha)
def test_one_pole_numerator_denominator_constructor(al, a):      # :226-233
  filt = al.ZFilter(numerator=[1.], denominator=[1., -a])
  expected = [x for x in data]
  for i in range(1, len(expected)):
    expected[i] += a * expected[i - 1]
  assert almost_eq(list(filt(data)), expected)


@p("delay", range(1, 5))
def test_one_delay_variable_gain(al, delay):                     # :296-308
  gain = al.Stream(itertools.cycle(alpha))
  filt = gain * al.z ** -delay
  assert isinstance(filt, al.ZFilter)
  length = 50
  padded = itertools.chain([0.] * delay, itertools.cycle(data))
  expected = [g * d for g, d in itertools.islice(zip(itertools.cycle(alpha), padded), length)]
  result = filt(itertools.cycle(data))
  assert isinstance(result, al.Stream)
  assert almost_eq(result.take(length), expected)


def test_variable_gain_in_denominator(al):                       # :310-330
  z, Stream = al.z, al.Stream
  filt = 1 / (Stream(1, 2, 3) - z ** -1)
  assert isinstance(filt, al.ZFilter)
  ainv = [1, .5, 1. / 3]
  expected_filt1 = Stream(ainv) / (1 - Stream(ainv) * z ** -1)
  assert isinstance(expected_filt1, al.ZFilter)
  r = list(filt(itertools.cycle(data)))                  # the 3-item gain Stream ends the output
  ex1 = list(expected_filt1(itertools.cycle(data)))
  assert len(r) == 3 and almost_eq(r, ex1)
  y, out = 0., []                                        # y[n] = (x[n] + y[n-1]) / a[n], by hand
  for a, x in zip([1, 2, 3], data):
    y = (x + y) / a
    out.append(y)
  assert almost_eq(r, out)


def test_readme_lpc_example_filters(al):                         # README.rst:362-373 (filter part)
  z = al.z
  blk = [-1., 0., 1., 0.] * 50
  analysis_filt = 1 + 0.5 * z ** -2 - 0.5 * z ** -4      # what lpc.covar(blk, 4) returns there
  residual = list(analysis_filt(blk))
  assert residual[:10] == [-1.0, 0.0, 0.5, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
  synth_filt = 1 / analysis_filt
  assert synth_filt(residual).take(10) == [-1.0, 0.0, 1.0, 0.0, -1.0, 0.0, 1.0, 0.0, -1.0, 0.0]
