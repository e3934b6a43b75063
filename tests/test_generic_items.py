"""Items the float64 engine does not take -- all-integer calls, complex numbers, Fractions, NumPy matrices with
matrix-valued coefficient Streams -- run on audiolazy_amd/generic.py, the product's own per-sample path with the
reference's semantics (SURVEY.md 8b accept gate: "else fall back to a per-sample Python generator").  Golden
values: tests/golden/generic_items.json, produced by the reference itself (oracle/gen_golden.py, ``repr`` strings).
No GPU involved: these calls never reach the engine."""
import itertools
from fractions import Fraction
from math import cos, pi, sqrt

import numpy as np
import pytest

from conftest import load_golden


pytestmark = pytest.mark.filterwarnings("ignore::PendingDeprecationWarning")


def golden():
  return {c["tag"]: c for c in load_golden("generic_items.json")}


def check(tag, result):
  want = golden()[tag]
  items = list(result)
  assert [type(v).__name__ for v in items] == want["types"], tag
  assert [repr(v.tolist()) if hasattr(v, "tolist") else repr(v) for v in items] == want["reprs"], tag


def test_all_integer_calls_keep_integers():
  from audiolazy_amd import ZFilter, z
  filt = ZFilter([1, 1], [1, -1])
  check("int_doctest", filt([1, 5, -4, -7, 9], memory=[3], zero=0))                 # lazy_filters.py:735-742
  check("int_doctest_delayed", (filt * z ** -1)([4, 10, 11, 0, 2], zero=0))
  check("int_gain_only", ZFilter([3])([1, 2, -5]))                                   # ``zero`` is never read
  check("int_then_float_items", ZFilter([3, 1])([1, 2.5, 2, 7], zero=0))
  check("int_division_by_gain", ZFilter([1, 1], [2, -1])([1, 5, -4, -7, 9], zero=0))
  check("int_negative_gain", ZFilter([2, 1], [-1, 3])([1, 5, -4], zero=0))
  check("fraction_items", ZFilter([1, 2], [1, -1])([Fraction(1, 3), Fraction(2, 7), Fraction(-5, 2)], zero=0))


def test_complex_items_and_coefficients():
  from audiolazy_amd import Stream, ZFilter, z
  check("complex_items", (1 - z ** -1)([1j, 2, 3 + 1j, -1.5j]))
  check("complex_coefficients", ZFilter([1, .5j], [1, -.25 + .1j])([1j, 2, 3 + 1j, -1.5j, 0, 1]))
  check("complex_series", (Stream(itertools.cycle([1j, 2.])) + z ** -1)([1., 2., 3., 4.]))


def test_constants_travel_as_text():
  """The reference formats constant coefficients, the gain and a term-less filter's ``zero`` into generated source
  (lazy_filters.py:209, 224, 229-231, 236): what is not its own literal is re-read by the parser.  Values AND types
  from the reference (round 5; all of these stay on the per-sample path)."""
  from decimal import Decimal
  from audiolazy_amd import ZFilter
  check("text_fraction_coefficients", ZFilter([Fraction(3, 5), 1], [1, Fraction(-1, 4)])([1, 2, 3], zero=0))
  check("text_fraction_gain", ZFilter([1, 2], [Fraction(3, 2), -1])([1, 2, 3], zero=0))
  check("text_fraction_coefficient_fraction_items", ZFilter([Fraction(3, 5)], [1])([Fraction(1, 3), Fraction(2, 7)], zero=0))
  check("text_negative_fraction_denominator", ZFilter([1], [1, Fraction(-3, 7), Fraction(2, 9)])([1, 2, 3, 4], zero=0))
  check("text_zero_fraction_whole", ZFilter([0], [1])([1, 2], zero=Fraction(3)))
  check("text_zero_fraction", ZFilter([0], [1])([1, 2], zero=Fraction(-5, 4)))
  check("text_zero_complex", ZFilter([0], [1])([1, 2], zero=-2j))
  check("text_gain_minus_1j", ZFilter([1, 1], [-1j, .5])([1., 2., 3.]))
  check("text_gain_2j", ZFilter([1, 1], [2j, .5])([1., 2., 3.]))
  check("text_complex_denominator", ZFilter([1], [1, -2j])([1., 2., 3.]))
  check("text_bool_coefficient", ZFilter([True, 2], [1])([1, 2, 3], zero=0))
  check("text_numpy_int_coefficient", ZFilter([np.int64(3), 2], [1])([1, 2, 3], zero=0))
  check("text_numpy_complex_coefficient", ZFilter([np.complex128(1 + 2j), 2], [1])([1, 2, 3], zero=0))
  check("text_numpy_float_coefficient_complex_items", ZFilter([np.float64(.1), 2], [1, np.float64(-.5)])([1j, 2, 3 - 1j]))
  check("text_decimal_coefficient", ZFilter([Decimal("0.1"), 2], [1])([1, 2, 3], zero=0))


def test_matrix_items_with_matrix_coefficient_streams():
  """tests/test_filters_extdep.py:49-89 of the reference: 2x2 matrix coefficient Streams on a 2x3 matrix signal."""
  from audiolazy_amd import Stream, z
  mat = np.matrix
  rep = lambda v: Stream(itertools.repeat(v))
  m, n1, n2 = mat([[1, 2], [2, 2]]), mat([[1.2, 3.2], [1.2, 1.1]]), mat([[-1, 2], [-1, 2]])
  a = mat([[.3, .4], [.5, .6]])
  filt = (rep(m) + Stream(itertools.cycle([n1, n2])) * z ** -1) / (1 - rep(a) * z ** -1)
  data = [itertools.cycle([1, 2]), itertools.count(), itertools.count(1, 2), itertools.cycle([.2, .33, .77, pi, cos(3)]),
          itertools.repeat(pi), (sqrt(2) + k * (pi / 3) for k in itertools.count())]
  sig = (mat(vect).reshape(2, 3) for vect in zip(*data))
  res = filt(sig, zero=mat([[0, 0, 0], [0, 0, 0]]))
  want = golden()["matrix_items_matrix_series"]
  got = list(itertools.islice(res, 12))
  assert all(type(v).__name__ == "matrix" for v in got)
  for v, r in zip(got, want["reprs"]):
    # (count(start=sqrt(2), step=pi/3) accumulates in the reference; the closed form above may differ in the last bits)
    np.testing.assert_allclose(np.asarray(v), np.asarray(eval(r)), rtol=1e-13)
  # the reference's own check of this case: y = m x + n x' + a y'
  xs = list(itertools.islice((mat(vect).reshape(2, 3) for vect in zip(
      itertools.cycle([1, 2]), itertools.count(), itertools.count(1, 2), itertools.cycle([.2, .33, .77, pi, cos(3)]),
      itertools.repeat(pi), (sqrt(2) + k * (pi / 3) for k in itertools.count()))), 12))
  old_x = old_y = mat(np.zeros((2, 3)))
  for k, (x, y) in enumerate(zip(xs, got)):
    exp = m * x + (n1 if k % 2 == 0 else n2) * old_x + a * old_y
    np.testing.assert_allclose(np.asarray(y), np.asarray(exp), rtol=1e-12)
    old_x, old_y = x, exp


def test_the_gate_itself():
  from audiolazy_amd import generic
  assert generic.all_int_configuration([1, 1], [1, -1], [3], 0)
  assert not generic.all_int_configuration([1, 1], [1, -1], None, 0.)        # zero is read (delay tap, memory fill)
  assert generic.all_int_configuration([3], [1], None, 0.)                    # ... and here it is not
  assert generic.all_int_configuration([0.0, 1, 1], [1, -1], None, 0)         # the dense lists pad with 0.0
  assert not generic.all_int_configuration([1., 1], [1, -1], None, 0)
  assert not generic.all_int_configuration([1, 1], [1, -1], [3.5], 0)
  assert generic.coefficients_fit_engine([1, 2.5, np.float64(3)], [1]) and not generic.coefficients_fit_engine([1j], [1])
  assert generic.is_engine_item(1) and generic.is_engine_item(2.5) and generic.is_engine_item(np.zeros(4))
  assert generic.is_engine_item([1., 2.]) and not generic.is_engine_item(1j) and not generic.is_engine_item(np.matrix([[1.]]))
  assert not generic.is_engine_item(Fraction(1, 2)) and not generic.is_engine_item(np.zeros((2, 2)))
  assert generic.initial_memory([.7], 2, 0.) == [0., .7]                      # LEFT-padded (lazy_filters.py:193-195)
  assert generic.initial_memory(lambda n: range(n), 3, 0.) == [0, 1, 2] and generic.initial_memory(None, 2, 5) == [5, 5]
  with pytest.raises(ZeroDivisionError):
    list(generic.df1([1], [0, 1], [1, 2]))
  assert list(generic.df1([0], [1], [1, 2, 3], zero=7)) == [7, 7, 7]           # no terms: ``zero`` per item (:227-231)


def test_product_never_imports_the_oracle():
  import os
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  src = open(os.path.join(root, "audiolazy_amd", "generic.py")).read()
  assert "import oracle" not in src and "from oracle" not in src and "/root/reference" not in src


def test_memory_is_read_once_at_call_time():
  """``memory`` as a one-shot iterator, a Stream or a callable (reference lazy_filters.py:185-195): read once, when
  the filter is CALLED; the gate never draws from it (round-3 advisor: it did, and an integer-coefficient filter
  lost its initial state).  Containers hand the same object to every member in turn (:988-990, :1052-1054)."""
  from audiolazy_amd import CascadeFilter, ParallelFilter, Stream, ZFilter, generic
  from audiolazy_amd.bank import stage_memories
  acc, two = ZFilter([1, 1], [1, -1]), ZFilter([1], [1, 0, -1])
  data = [1, 5, -4, -7, 9]
  check("mem_iter_int", acc(data, memory=iter([3]), zero=0))
  check("mem_stream_int", acc(data, memory=Stream([3, 8]), zero=0))
  check("mem_iter_short_int", two(data, memory=iter([3]), zero=0))
  check("mem_callable_int", two(data, memory=lambda n: [7] * n, zero=0))
  check("mem_cascade_iter_int", CascadeFilter(acc, two, acc)(data, memory=iter([3, 4, 5, 6, 7, 8, 9]), zero=0))
  check("mem_cascade_list_int", CascadeFilter(acc, two, acc)(data, memory=[3, 4, 5], zero=0))
  check("mem_parallel_iter_int", ParallelFilter(acc, two, acc)(data, memory=iter([3, 4, 5, 6, 7, 8, 9]), zero=0))
  # the advisor's reproduction: the gate leaves a one-shot iterator alone
  mem = iter([5.0])
  assert generic.all_int_configuration([1, 1], [1, -1], mem, 0.) is False
  assert stage_memories(mem, [2]) == [[5.0]]
  # staged at call time: what happens to the caller's object afterwards does not matter
  pulled = []

  def memory_source():
    for v in (3, 4):
      pulled.append(v)
      yield v
  res = acc(data, memory=memory_source(), zero=0)
  assert pulled == [3, 4]          # (the reference's takewhile draws the item that ends it too)
  assert list(res) == [4, 10, 11, 0, 2]
  assert generic.read_memory(None, 3, 0.) is None and generic.read_memory([.7], 2, 0.) == [0., .7]
  assert generic.read_memory(iter(range(9)), 2, 0.) == [0, 1] and generic.read_memory(lambda n: [1.] * n, 2, 0) == [1., 1.]
