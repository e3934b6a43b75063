"""The algebra-only tests of the reference's TestZFilter (audiolazy/tests/test_filters.py:120-262),
restated with audiolazy_amd in place of audiolazy.  No filter is called here, so no GPU."""
import operator

import pytest

from audiolazy_amd import z, ZFilter

p = pytest.mark.parametrize
alpha = [-.5, -.2, -.1, 0, .1, .2, .5]


def almost_eq(a, b):
  """The reference's tolerance (lazy_misc.py:264-267): |a - b| <= 2**-23 * |a + b| item by item."""
  a, b = list(a), list(b)
  if len(a) != len(b):
    return False
  for u, v in zip(a, b):
    if isinstance(u, tuple):
      if not almost_eq(u, v):
        return False
    elif abs(u - v) > 2 ** -23 * abs(u + v):
      return False
  return True


@p("a", alpha)
@p("b", alpha)
def test_z_division(a, b):                                       # :117-133
  idx_den1, idx_num2 = -1, -2
  for idx_num1 in range(-3, 1):
    for idx_den2 in range(-18, 1, 6):
      fa, fb, fc, fd = (a * z ** idx_num1, 2 + b * z ** idx_den1, 3 * z ** idx_num2, 1 + 5 * z ** idx_den2)
      my_filt = (fa / fb) / (fc / fd)
      idx_corr = max(idx_num2, idx_num2 + idx_den1)
      num_filter = fa * fd * (z ** -idx_corr)
      den_filter = fb * fc * (z ** -idx_corr)
      assert almost_eq(my_filt.numpoly.terms(), num_filter.numpoly.terms())
      assert almost_eq(my_filt.denpoly.terms(), den_filter.numpoly.terms())


@p("a", alpha)
def test_z_grouped_powers(a):                                    # :172-199
  base = 1 + a * z ** -1
  f1, f2, f3, f4 = base ** -1, 3 * base ** -2, base ** 2, a * base ** 0
  f5 = (base ** 3) * (base ** -4)
  f6 = ((1 - a * z ** -1) / base ** 2) ** 2
  one = [1.]
  assert almost_eq(f1.numerator, one) and almost_eq(f1.denominator, [1., a] if a != 0 else one)
  assert almost_eq(f2.numerator, [3.]) and almost_eq(f2.denominator, [1., 2 * a, a * a] if a != 0 else one)
  assert almost_eq(f3.numerator, [1., 2 * a, a * a] if a != 0 else one) and almost_eq(f3.denominator, one)
  assert almost_eq(f4.numerator, [a] if a != 0 else []) and almost_eq(f4.denominator, one)
  assert almost_eq(f5.numerator, [1., 3 * a, 3 * a * a, a * a * a] if a != 0 else one)
  assert almost_eq(f5.denominator, [1., 4 * a, 6 * a * a, 4 * a * a * a, a * a * a * a] if a != 0 else one)
  assert almost_eq(f6.numerator, [1., -2 * a, a * a] if a != 0 else one)
  assert almost_eq(f6.denominator, [1., 4 * a, 6 * a * a, 4 * a * a * a, a * a * a * a] if a != 0 else one)


@p("a", alpha)
def test_z_one_pole_added_one_pole(a):                           # :211-218
  filt = -(3 / (1 + a * z ** -1)) + (2 / (1 - a * z ** -1))
  assert almost_eq(filt.numerator, [-1., 5 * a] if a != 0 else [-1.])
  assert almost_eq(filt.denominator, [1., 0., -a * a] if a != 0 else [1.])


@p("a", alpha)
def test_z_one_pole_added_to_a_number(a):                        # :220-224
  filt = -(5 / (1 - a * z ** -1)) + a
  assert almost_eq(filt.numerator, [-5 + a, -a * a] if a != 0 else [-5])
  assert almost_eq(filt.denominator, [1., -a] if a != 0 else [1.])


@p("delay", range(1, 7))
def test_diff_twice_only_numerator_one_delay(delay):             # :235-246
  data = z ** -delay
  ddz = data.diff()
  assert almost_eq(ddz.numerator, [0] * delay + [0, -delay]) and almost_eq(ddz.denominator, [1])
  ddz2 = ddz.diff()
  assert almost_eq(ddz2.numerator, [0] * delay + [0, 0, delay * (delay + 1)]) and almost_eq(ddz2.denominator, [1])
  alt = data.diff(2)
  assert almost_eq(ddz2.numerator, alt.numerator) and almost_eq(ddz2.denominator, alt.denominator)


def test_diff():                                                 # :248-252
  ddz = ((1 + z ** -1) / (1 - z ** -1)).diff()
  assert almost_eq(ddz.numerator, [0, 0, -2]) and almost_eq(ddz.denominator, [1, -2, 1])


def test_variable_gain_builds_filters_not_streams():             # :296-304, :306-313 (types)
  from audiolazy_amd import Stream
  filt = Stream([.1, .2, .3]) * z ** -2
  assert isinstance(filt, ZFilter) and not filt.is_lti()
  filt = 1 / (Stream(1, 2, 3) - z ** -1)
  assert isinstance(filt, ZFilter) and not filt.is_lti()
