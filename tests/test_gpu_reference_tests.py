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


delays = list(range(1, 5))


@p("delay", delays)
def test_one_delay_variable_gain(al, delay):                     # :296-308
  cycle, z, zero_pad = al.cycle, al.z, al.zero_pad
  gain = cycle(alpha)
  filt = gain * z ** -delay
  length = 50
  assert isinstance(filt, al.ZFilter)
  data_stream = cycle(alpha) * zero_pad(cycle(data), left=delay)
  expected = data_stream.take(length)
  result_stream = filt(cycle(data))
  assert isinstance(result_stream, al.Stream)
  assert almost_eq(result_stream.take(length), expected)


def test_variable_gain_in_denominator(al):                       # :310-330
  Stream, z, cycle, thub = al.Stream, al.z, al.cycle, al.thub
  a = Stream(1, 2, 3)
  filt = 1 / (a - z ** -1)
  assert isinstance(filt, al.ZFilter)
  ainv = Stream(1, .5, 1. / 3)
  expected_filt1 = ainv.copy() / (1 - ainv.copy() * z ** -1)
  assert isinstance(expected_filt1, al.ZFilter)
  ai = thub(ainv, 2)
  expected_filt2 = ai / (1 - ai * z ** -1)
  assert isinstance(expected_filt2, al.ZFilter)
  length = 50
  expected1, expected2 = expected_filt1(cycle(data)), expected_filt2(cycle(data))
  result = filt(cycle(data))
  assert all(isinstance(s, Stream) for s in (expected1, expected2, result))
  r, ex1, ex2 = result.take(length), expected1.take(length), expected2.take(length)
  assert almost_eq(r, ex1) and almost_eq(r, ex2) and almost_eq(ex1, ex2)


@p("delay", delays)
def test_fir_time_variant_sum(al, delay):                        # :332-347
  cycle, z, zero_pad = al.cycle, al.z, al.zero_pad
  gain1, gain2 = cycle(alpha), cycle(alpha[-2::-1])
  filt = gain1 * z ** -delay + gain2 * z ** -delays[0]
  length = 50
  assert isinstance(filt, al.ZFilter)
  data_stream1 = cycle(alpha) * zero_pad(cycle(data), left=delay)
  data_stream2 = cycle(alpha[-2::-1]) * zero_pad(cycle(data), left=delays[0])
  expected = data_stream1 + data_stream2
  result = filt(cycle(data))
  assert isinstance(expected, al.Stream) and isinstance(result, al.Stream)
  assert almost_eq(result.take(length), expected.take(length))


@p("delay", delays)
def test_iir_time_variant_sum(al, delay):                        # :349-392
  Stream, z, cycle = al.Stream, al.z, al.cycle
  gain1, gain2 = cycle(alpha), cycle(alpha[-2::-1])
  gain3, gain4, gain5, gain6 = Stream(1, 2, 3), Stream(.1, .7, -.5, -1e-3), Stream(.1, .2), Stream(3, 2, 1, 0)
  num1 = gain1.copy() * z ** -delay + gain2.copy() * z ** -delays[0]
  den1 = 1 + gain3.copy() * z ** -(delay + 2)
  filt1 = num1 / den1
  num2 = gain4.copy() * z ** -delay + gain5.copy() * z ** -delays[-1]
  den2 = 1 + gain6.copy() * z ** -(delay - 1)
  filt2 = num2 / den2
  filt = filt1 + filt2
  assert all(isinstance(f, al.ZFilter) for f in (num1, den1, filt1, num2, den2, filt2, filt))
  length = 90
  expected_filter = (
    gain1.copy() * z ** -delay +
    gain2.copy() * z ** -delays[0] +
    gain1.copy() * gain6.copy() * z ** -(2 * delay - 1) +
    gain2.copy() * gain6.copy() * z ** -(delay + delays[0] - 1) +
    gain4.copy() * z ** -delay +
    gain5.copy() * z ** -delays[-1] +
    gain4.copy() * gain3.copy() * z ** -(2 * delay + 2) +
    gain5.copy() * gain3.copy() * z ** -(delay + delays[-1] + 2)
  ) / (
    1 +
    gain3.copy() * z ** -(delay + 2) +
    gain6.copy() * z ** -(delay - 1) +
    gain3.copy() * gain6.copy() * z ** -(2 * delay + 1)
  )
  assert isinstance(expected_filter, al.ZFilter)
  expected, result = expected_filter(cycle(data)), filt(cycle(data))
  assert isinstance(expected, Stream) and isinstance(result, Stream)
  assert almost_eq(result.take(length), expected.take(length))


@p("delay", delays)
def test_fir_time_variant_multiplication(al, delay):             # :394-415
  z, cycle = al.z, al.cycle
  gain1, gain2, gain3 = cycle(alpha), cycle(alpha[::2]), 2 + cycle(alpha[::3])
  filt1 = gain1.copy() * z ** -delay + gain2.copy() * z ** -(delay + 1)
  filt2 = gain2.copy() * z ** -delay + gain3.copy() * z ** -(delay - 1)
  filt = filt1 * filt2
  expected_filter = (
    (gain1.copy() + gain3.copy()) * gain2.copy() * z ** -(2 * delay) +
    gain1.copy() * gain3.copy() * z ** -(2 * delay - 1) +
    gain2.copy() ** 2 * z ** -(2 * delay + 1)
  )
  length = 80
  assert all(isinstance(f, al.ZFilter) for f in (filt1, filt2, expected_filter))
  expected, result = expected_filter(cycle(data)), filt(cycle(data))
  assert almost_eq(result.take(length), expected.take(length))


@p("delay", delays)
def test_iir_time_variant_multiplication(al, delay):             # :417-458
  Stream, z, cycle = al.Stream, al.z, al.cycle
  gain1, gain2 = cycle([4, 5, 6, 5, 4, 3]), cycle(alpha[::-1])
  gain3, gain4 = Stream(*(alpha + [1, 2, 3])), Stream(.1, -.2, .3)
  gain5, gain6 = Stream(.1, .1, .1, -7), Stream(3, 2)
  num1 = gain1.copy() * z ** -delay + gain2.copy() * z ** -delays[0]
  den1 = 1 + gain3.copy() * z ** -(delay - 1)
  filt1 = num1 / den1
  num2 = gain4.copy() * z ** -delay + gain5.copy() * z ** -delays[-1]
  den2 = 1 + gain6.copy() * z ** -(delay + 5)
  filt2 = num2 / den2
  filt = filt1 * filt2
  assert all(isinstance(f, al.ZFilter) for f in (num1, den1, filt1, num2, den2, filt2, filt))
  length = 90
  expected_filter = (
    gain1.copy() * gain4.copy() * z ** -(2 * delay) +
    gain2.copy() * gain4.copy() * z ** -(delay + delays[0]) +
    gain1.copy() * gain5.copy() * z ** -(delay + delays[-1]) +
    gain2.copy() * gain5.copy() * z ** -(delays[0] + delays[-1])
  ) / (
    1 +
    gain3.copy() * z ** -(delay - 1) +
    gain6.copy() * z ** -(delay + 5) +
    gain3.copy() * gain6.copy() * z ** -(2 * delay + 4)
  )
  assert isinstance(expected_filter, al.ZFilter)
  expected, result = expected_filter(cycle(data)), filt(cycle(data))
  assert almost_eq(result.take(length), expected.take(length))


def test_copy(al):                                               # :460-473
  Stream, z, cycle = al.Stream, al.z, al.cycle
  filt1 = (2 + Stream(1, 2, 3) * z ** -1) / Stream(1, 5)
  filt2 = filt1.copy()
  assert isinstance(filt2, al.ZFilter) and filt1 is not filt2
  filt_ex = Stream(2, .4) + Stream(1, .4, 3, .2, 2, .6) * z ** -1
  length = 50
  r1 = filt1(cycle(data[::-1])).take(length)
  r2 = filt2(cycle(data[::-1])).take(length)
  ex = filt_ex(cycle(data[::-1])).take(length)
  assert almost_eq(r1, ex) and almost_eq(r2, ex) and almost_eq(r1, r2)


@p("delay", delays)
def test_iir_time_variant_sum_with_copy(al, delay):              # :475-494
  Stream, z, cycle, thub = al.Stream, al.z, al.cycle, al.thub
  k = thub(min(alpha) + 2 + cycle(alpha), 3)
  filt = z ** -2 / k + Stream(5, 7) * z / (1 + z ** -delay)
  filt += filt.copy()
  filt *= z ** -1
  assert isinstance(filt, al.ZFilter)
  length = 40
  expected_filter = (
    2 / k * z ** -3 +
    2 / k * z ** -(delay + 3) +
    Stream(10, 14)
  ) / (1 + z ** -delay)
  assert isinstance(expected_filter, al.ZFilter)
  expected, result = expected_filter(cycle(data)), filt(cycle(data))
  assert almost_eq(result.take(length), expected.take(length))


def test_hashable(al):                                           # :496-502
  z = al.z
  filt = 1 / (7 + z ** -1)
  my_set = {filt, 17, z, z ** -1, object}
  assert z in my_set and z ** -1 in my_set and filt in my_set and -z not in my_set


# ------------------------------------------------------------- containers (:505-566)
@p("kind", ["CascadeFilter", "ParallelFilter"])
def test_container_add_mul(al, kind):                            # :508-520
  cls, z = getattr(al, kind), al.z
  filt_sum = cls(z) + cls(z + 3)
  assert isinstance(filt_sum, cls) and filt_sum == cls(z, z + 3)
  filt_prod = cls(1 - z ** -1) * 3
  assert isinstance(filt_prod, cls) and filt_prod == cls(1 - z ** -1, 1 - z ** -1, 1 - z ** -1)


@p("kind", ["CascadeFilter", "ParallelFilter"])
def test_container_non_linear(al, kind):                         # :522-536
  import math
  cls, z = getattr(al, kind), al.z
  for filts in ((lambda d: d ** 2), (z ** -1, lambda d: d + 4), (1 / z ** -2, (lambda d: 0.), z ** -1)):
    filt = cls(filts)
    assert isinstance(filt, cls) and not filt.is_linear()
    for attr in ("numpoly", "denpoly"):
      with pytest.raises(AttributeError):
        getattr(filt, attr)
    with pytest.raises(AttributeError):
      filt.freq_response(math.pi / 2)


@p("kind", ["CascadeFilter", "ParallelFilter"])
def test_container_const_filter(al, kind):                       # :538-548
  import functools
  cls = getattr(al, kind)
  values = [2, 4, 3, 7 - 1, -8]
  filt1, filt2 = cls(*values), cls(values)
  func = operator.mul if kind == "CascadeFilter" else operator.add
  expected_value = functools.reduce(func, values)
  count = 10
  for d in values:
    expected = [d * expected_value] * count
    assert filt1(al.Stream(d)).take(count) == expected
    assert filt2(al.Stream(d)).take(count) == expected


@p("source", [list(range(3)), "stream", [.2, .5, .4, .1]])
def test_call_empty_containers(al, source):                      # :559-566
  mk = (lambda: al.Stream([5., 4., 6., 7., 12., -2.])) if source == "stream" else (lambda: source)
  ref = [5., 4., 6., 7., 12., -2.] if source == "stream" else source
  assert list(al.CascadeFilter()(mk())) == list(ref)
  assert all(el == 0. for el in al.ParallelFilter()(mk()))


def test_container_with_a_plain_callable_member(al):
  """A non-linear member runs as the plain composition / sum of the members' own calls."""
  z = al.z
  x = [1., 2., 3., 4.]
  casc = al.CascadeFilter(z ** -1, lambda d: al.Stream(d) + 4)
  assert list(casc(x)) == [4., 5., 6., 7.]
  par = al.ParallelFilter(z ** -1, lambda d: al.Stream(d) * 2)
  assert list(par(x)) == [2., 5., 8., 11.]


def test_readme_lpc_example_filters(al):                         # README.rst:362-373 (filter part)
  z = al.z
  blk = [-1., 0., 1., 0.] * 50
  analysis_filt = 1 + 0.5 * z ** -2 - 0.5 * z ** -4      # what lpc.covar(blk, 4) returns there
  residual = list(analysis_filt(blk))
  assert residual[:10] == [-1.0, 0.0, 0.5, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
  synth_filt = 1 / analysis_filt
  assert synth_filt(residual).take(10) == [-1.0, 0.0, 1.0, 0.0, -1.0, 0.0, 1.0, 0.0, -1.0, 0.0]


# ------------------------------------------------------------------------------------------
# The same expressions against what the REFERENCE computes for them (tests/golden/
# timevar_algebra.json, generated by oracle/gen_golden.py): the Stream-coefficient arithmetic of
# Poly / ZFilter has to perform the reference's operations in the reference's order.
def _same_bits(a, b):
  import numpy as np
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return a.shape == b.shape and bool(np.all(a.view(np.uint64) == b.view(np.uint64)))


def test_time_variant_algebra_matches_the_reference_bit_for_bit(al):
  from conftest import load_golden, unhex
  Stream, z, cycle, thub = al.Stream, al.z, al.cycle, al.thub
  for case in load_golden("timevar_algebra.json"):
    delay, name = case["delay"], case["name"]
    if name == "iir_sum":
      gain1, gain2 = cycle(alpha), cycle(alpha[-2::-1])
      gain3, gain4, gain5, gain6 = Stream(1, 2, 3), Stream(.1, .7, -.5, -1e-3), Stream(.1, .2), Stream(3, 2, 1, 0)
      filt1 = (gain1.copy() * z ** -delay + gain2.copy() * z ** -delays[0]) / (1 + gain3.copy() * z ** -(delay + 2))
      filt2 = (gain4.copy() * z ** -delay + gain5.copy() * z ** -delays[-1]) / (1 + gain6.copy() * z ** -(delay - 1))
      got = (filt1 + filt2)(cycle(data)).take(90)
    elif name == "iir_mul":
      gain1, gain2 = cycle([4, 5, 6, 5, 4, 3]), cycle(alpha[::-1])
      gain3, gain4 = Stream(*(alpha + [1, 2, 3])), Stream(.1, -.2, .3)
      gain5, gain6 = Stream(.1, .1, .1, -7), Stream(3, 2)
      filt1 = (gain1.copy() * z ** -delay + gain2.copy() * z ** -delays[0]) / (1 + gain3.copy() * z ** -(delay - 1))
      filt2 = (gain4.copy() * z ** -delay + gain5.copy() * z ** -delays[-1]) / (1 + gain6.copy() * z ** -(delay + 5))
      got = (filt1 * filt2)(cycle(data)).take(90)
    elif name == "copy":
      got = ((2 + Stream(1, 2, 3) * z ** -1) / Stream(1, 5)).copy()(cycle(data[::-1])).take(50)
    elif name == "sum_with_copy":
      k = thub(min(alpha) + 2 + cycle(alpha), 3)
      filt = z ** -2 / k + Stream(5, 7) * z / (1 + z ** -delay)
      filt += filt.copy()
      filt *= z ** -1
      got = filt(cycle(data)).take(40)
    else:
      got = (1 / (Stream(1, 2, 3) - z ** -1))(cycle(data)).take(50)
    assert _same_bits(got, unhex(case["y"])), (name, delay)
