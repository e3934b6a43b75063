"""A subset of the reference's TestPoly (audiolazy/tests/test_poly.py) restated with
audiolazy_amd.Poly: the parts of the polynomial algebra the z-algebra of the filter path rests
on (construction, + - * /, values / order, diff, equality, Stream coefficients).  The reference's
``zero=`` keyword, ``integrate``, ``__setitem__``, ``roots``, ``lagrange`` and ``resample`` are
not mirrored (not on the filter path)."""
import itertools
import operator

import pytest

from audiolazy_amd import Poly, x, Stream
from test_reference_algebra import almost_eq

p = pytest.mark.parametrize

example_data = [[1, 2, 3], [-7, 0, 3, 0, 5], [1], list(range(2, -6, -1))]
instances = [Poly([1.7, 2, 3.3]), Poly({-2: 1, -1: 5.1, 3: 2}), Poly({-1.1: 1, 1.1: .5})]
polynomials = [12 * x ** 2 + .5 * x + 18, .45 * x ** 17 + 2 * x ** 5 - x + 8, 8 * x ** 5 + .2 * x ** 3 + .1 * x ** 2,
               42.7 * x ** 4, 8 * x ** 3 + 3 * x ** 2 + 22.2 * x + .17]
diff_table = [(x + 2, 1), (Poly({0: 22}), 0), (Poly({}), 0),
              (x ** 2 + 2 * x + x ** -7 + x ** -.2 - 4, 2 * x + 2 - 7 * x ** -8 - .2 * x ** -1.2)]


@p("data", example_data)
def test_len_iter_from_list(data):                               # :76-80
  assert len(Poly(data)) == len([k for k in data if k != 0])
  assert list(Poly(data).values()) == data


def test_empty():                                                # :82-86
  assert len(Poly()) == 0 and not Poly() and len(Poly([])) == 0 and not Poly([])


@p(("key", "value"), [(3, 2), (7, 12), (500, 8), (0, 12), (1, 8)])
def test_input_dict_one_item(key, value):                        # :88-91
  assert list(Poly({key: value}).values()) == [0] * key + [value]


def test_input_dict_three_items_and_fake_zero():                 # :93-97
  polynomial = Poly({8: 5, 7: -1, 6: 80, 9: 0})
  assert len(polynomial) == 3 and list(polynomial.values()) == [0] * 6 + [80, -1, 5]


@p("data", example_data)
def test_output_dict(data):                                      # :99-102
  assert dict(Poly(data).terms()) == {k: v for k, v in enumerate(data) if v != 0}


def test_sum_and_float_sub():                                    # :104-112
  assert Poly([2, 3, 4]) + Poly([0, 5, -4, -1]) == Poly([2, 8, 0, -1])
  poly_obj = Poly([.3, 4]) - Poly() - Poly([0, 4, -4]) + Poly([.7, 0, -4])
  assert len(poly_obj) == 1 and abs(poly_obj[0] - 1) <= 2 ** -23 * 2


@p("val", instances)
@p("den", [.1, -4e-3, 2])
def test_int_float_div(val, den):                                # :114-125
  div = operator.truediv
  assert almost_eq(div(den * val, den).terms(), val.terms())
  assert almost_eq(div(den * val, -den).terms(), (-val).terms())
  assert almost_eq(div(den * -val, den).terms(), (-val).terms())
  assert almost_eq(div(-den * val, den).terms(), (-val).terms())
  expected = Poly({k: v / den for k, v in val.terms()})
  assert almost_eq(div(val, den).terms(), expected.terms())
  assert almost_eq(div(val, -den).terms(), (-expected).terms())
  assert almost_eq(div(-val, den).terms(), (-expected).terms())
  assert almost_eq(div(-val, -den).terms(), expected.terms())


@p("poly", instances + polynomials)
def test_value_zero(poly):                                       # :127-133
  expected = ([v for k, v in poly.terms() if k == 0] + [0])[0]
  assert expected == poly(0) == poly(0.0) == poly[0] == poly[0.0]


@p("poly", polynomials)
def test_is_polynomial(poly):                                    # :139-149
  top = max(k for k, _ in poly.terms())
  assert poly.is_polynomial() and (poly + x ** 22.).is_polynomial() and (poly + 8).is_polynomial()
  assert (poly * x).is_polynomial() and (poly(-x) * .5).is_polynomial() and (poly * .5 * x ** 2).is_polynomial()
  assert not (poly * .5 * x ** .2).is_polynomial() and not (poly * x ** .5).is_polynomial()
  assert not (poly * x ** -(top + 1)).is_polynomial()


def test_values_order():                                         # :164-193
  poly = Poly({})
  assert list(poly.values()) == [] and poly.order == 0 and poly.is_polynomial()
  bad = Poly({-1: 3, 1: 2})
  with pytest.raises(AttributeError):
    bad.order
  assert not bad.is_polynomial()
  for poly in polynomials:
    order = max(k for k, _ in poly.terms())
    assert poly.order == order
    values = list(poly.values())
    for key, value in poly.terms():
      assert values[key] == value
      values[key] = 0
    assert values == [0] * (order + 1)


@p(("poly", "diff_poly"), diff_table)
def test_diff(poly, diff_poly):                                  # :215-217
  assert poly.diff() == diff_poly


def test_empty_comparison_to_zero_and_evaluation():              # :238-250
  inputs = [[], {}, [0, 0], [0], {25: 0}, {0: 0}, {-.2: 0}]
  values = [0, 0.] + [Poly(k) for k in inputs]
  for a, b in itertools.combinations_with_replacement(values, 2):
    assert a == b
  for data in inputs:
    poly = Poly(data)
    assert poly(5) == poly(0) == poly(-3) == poly(.2) == 0


def test_not_equal():                                            # :256-259
  for a, b in itertools.combinations(polynomials, 2):
    assert a != b


def test_pow_and_truediv_errors():                               # :302-324
  with pytest.raises(NotImplementedError):
    (x + 2) ** (.5 + x ** -1)
  with pytest.raises(NotImplementedError):
    (x ** 2 + 2) / (x + 1)
  with pytest.raises(ZeroDivisionError):
    (x + 2) / Poly()


@p("poly", polynomials)
def test_pow_basics(poly):                                       # :430-436
  assert poly ** 0 == 1 and poly ** Poly() == 1 and poly ** 1 == poly
  assert poly ** 2 == poly * poly and poly ** Poly(2) == poly * poly
  assert almost_eq((poly ** 3).terms(), (poly * poly * poly).terms())


def test_power_one_keep_integer():                               # :438-445
  for value in (0, -1, .5, 18):
    poly = Poly(1) ** value
    assert poly.order == 0 and poly[0] == 1 and isinstance(poly[0], int)


def test_hash_equalness_on_different_sorting():                  # :499-503
  a, b = Poly({1: 2, 5: 3}), Poly({5: 3, 1: 2})
  assert a == b and hash(a) == hash(b) and len({a, b}) == 1


def test_terms_order():                                          # :650-675
  for poly in polynomials:
    assert [k for k, _ in poly.terms()] == sorted(k for k, _ in poly.terms())
  laurent = Poly({-2: 1, 3: 2, 0: 5, -7: .5})
  assert [k for k, _ in laurent.terms(reverse=True)] == [3, 0, -2, -7]


# ------------------------------------------------------------- Stream coefficients (:353-470)
@p("data", [list(range(-7, 8)), [-3.2, 0, .5, 2.]])
@p("poly", polynomials)
def test_stream_evaluation(data, poly):                          # :352-357
  result = poly(Stream(data))
  assert isinstance(result, Stream)
  assert almost_eq(result, [poly(v) for v in data])


def test_stream_coeffs():                                        # :359-388
  poly = x * Stream(0, 2, 3) + 1
  assert isinstance(poly, Poly)
  result = poly(5)
  assert isinstance(result, Stream) and result.take(5) == [1, 11, 16, 1, 11]
  poly = x * Stream(0, 2, 3) + Stream(1, 8)
  assert poly(2).take(7) == [1, 12, 7, 8, 5, 14, 1]
  poly1 = x * Stream(0, 2, 3) + 1
  poly2 = x - Stream(1, 8)
  product = poly1 * poly2                       # (s1 x + 1)(x - s2) = s1 x^2 + (1 - s1 s2) x - s2
  assert isinstance(product, Poly)
  s1, s2 = Stream(0, 2, 3), Stream(1, 8)
  expected = (Stream(0, 2, 3) * 4 + (1 - s1 * s2) * 2 - Stream(1, 8)).take(9)
  assert product(2).take(9) == expected


def test_eq_ne_of_a_stream_copy():                               # :419-427
  poly = x * Stream(0, 1) + 1
  assert poly == poly and not poly != poly
  other = poly.copy()
  assert poly != other and not poly == other     # Streams compare by identity
