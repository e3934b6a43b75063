"""The reference's test_misc.py cases for the helpers mirrored here (rint, freq2lag / lag2freq,
almost_eq; audiolazy/tests/test_misc.py:35-167), restated with audiolazy_amd."""
import cmath
import itertools as it
import math

import pytest

from audiolazy_amd import rint, freq2lag, lag2freq, almost_eq, line, Stream

p = pytest.mark.parametrize
pi = math.pi
rint_table = [(.499, 0), (-.499, 0), (.5, 1), (-.5, -1), (1.00001e3, 1000), (-227.0090239, -227), (-12.95, -13)]


@p(("data", "expected"), rint_table)
def test_rint_default_step(data, expected):                      # :46-51
  result = rint(data)
  assert isinstance(result, int) and result == expected


@p(("data", "expected"), rint_table)
@p("n", [2, 3, 10])
def test_rint_step_n(data, expected, n):                         # :53-58
  result = rint(n * data, step=n)
  assert isinstance(result, int) and result == n * expected


def test_freq_lag_converters():                                  # :83-107
  for v in [37, 12, .5, -2, 1, .18, 4, 1e19, 2.7e-34]:
    assert freq2lag(v) == lag2freq(v)
    for a, b in it.permutations([lag2freq(freq2lag(v)), freq2lag(lag2freq(v)), v], 2):
      assert almost_eq(a, b)
  eq = 2.506628274631
  for k, v in {2.5: 2.5132741228718345, 30: 0.20943951023931953, 2: 3.141592653589793, eq: eq}.items():
    for f in (freq2lag, lag2freq):
      assert almost_eq(f(k), v) and almost_eq(f(v), k) and almost_eq(f(-k), -v) and almost_eq(f(-v), -k)


@p("aeq", list(almost_eq))
def test_almost_eq_single_and_complex_values(aeq):               # :112-136
  assert aeq(pi, pi) and aeq(0, 0) and aeq(0, 0.) and aeq(0., 0.) and aeq(18.7, 18.7) and aeq(15, 15)
  assert not aeq(15.0001, 15) and not aeq(1e-3, 1e-7) and not aeq(99999, 99999.03)
  assert not aeq(.99999, .9999903) and aeq(.99999, .9999901)
  assert not aeq(2j, 2) and not aeq(2j + 1, 2 + 1j) and not aeq(3 + 4j, 5)
  assert not aeq(3 + 4j, 3 + 4.0001j) and not aeq(3 + 4j, 2.99999 + 4j)
  assert aeq(3 + 4j, 1j + 3 * (1 + 1j)) and aeq(2j + 1, 2j + 1 + 1e-9 - 3e-8j)
  for a, b in line(28, 0, 2j * pi, finish=True).blocks(size=2, hop=1):
    assert not aeq(a, b)
    assert aeq(a, a * cmath.exp(2e-9j * pi)) and aeq(b, b * cmath.exp(-3e-9j * pi))


@p("aeq", list(almost_eq))
def test_almost_eq_iterables(aeq):                               # :138-158
  items = [1, 3, 2e-4, .5, .1, pi, 0, 0, 0., 12]
  assert aeq(items, Stream(items)) and aeq((d for d in sorted(items)), sorted(items))
  changed = items[:]
  changed[2] = 3j
  assert not aeq(changed, Stream(items))
  assert aeq([i * (1 + 2e-9) for i in items], Stream(items))
  assert aeq([], tuple()) and aeq(set(), tuple()) and aeq([], Stream([])) and aeq(([], []), [[], []])
  assert not aeq(([], [], []), [[], []]) and not aeq([[]], [])
  assert aeq(([], [], []), [[], []], pad=[])
  assert not aeq([], tuple(), ignore_type=False) and not aeq(set(), tuple(), ignore_type=False)
  assert not aeq([], Stream([]), ignore_type=False) and not aeq(([], []), [[], []], ignore_type=False)


def test_almost_eq_nested_iterables():                           # :160-167
  k = 1 + 1e-8
  items_list = [1, [3 + 1e-7, [2e-4 - 9e-14, .5]], [.1, pi * k], 0, [12]]
  items_tuple = [1 - 7e-8, (3, (2e-4, .5)), (.1, pi / k), 0., (12,)]
  assert almost_eq(items_list, items_tuple)
  items_list[-1][-1] = 11.9999
  assert not almost_eq(items_list, items_tuple)
