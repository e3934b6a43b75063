"""``filt(other_zfilter)`` -- substitution of a ZFilter for z -- and division by a zero-numerator filter, against
what the reference itself returns (reference lazy_filters.py:885-887 and :125-133).  Golden values:
tests/golden/composition.json, 640 seeded random rational pairs (constants, pure delays / advances, integer
coefficients, zero numerators) run through the unmodified reference by oracle/gen_golden.py.  Terms are compared
with their powers, values (bit for bit) AND Python types: the reference keeps ints ints."""
import pytest

from conftest import load_golden


def untyped(t):
  kind, v = t
  return int(v) if kind == "i" else float.fromhex(v)


def typed(v):
  assert isinstance(v, (int, float)) and not isinstance(v, bool), type(v)
  return ["i", int(v)] if isinstance(v, int) else ["f", float(v).hex()]


def outcome(fn):
  from audiolazy_amd import ZFilter
  try:
    r = fn()
  except Exception as exc:   # noqa: BLE001 -- the exception type is what is compared
    return dict(raises=type(exc).__name__)
  if not isinstance(r, ZFilter):
    return dict(value=typed(r))
  return dict(num=[[typed(k), typed(v)] for k, v in r.numpoly.terms()],
              den=[[typed(k), typed(v)] for k, v in r.denpoly.terms()])


def build(spec):
  from audiolazy_amd import ZFilter
  num, den = ({untyped(k): untyped(v) for k, v in side} for side in spec)
  return ZFilter(num, den)


CASES = load_golden("composition.json")


def test_fixture_has_the_families_the_review_asked_for():
  assert len(CASES) >= 400
  assert sum("raises" in c["f_of_g"] for c in CASES) >= 10          # zero denominators after substitution
  assert sum("raises" in c["g_over_f"] for c in CASES) >= 10         # division by a zero-numerator filter
  assert sum(all(t[1][0] == "i" for side in c["f"] for t in side) for c in CASES) >= 50   # all-integer filters


@pytest.mark.parametrize("key,expr", [
  ("f_built", lambda f, g: f()),
  ("g_built", lambda f, g: g()),
  ("f_of_g", lambda f, g: f()(g())),
  ("g_over_f", lambda f, g: g() / f()),
  ("k_over_f", lambda f, g: 2.5 / f()),
  ("f_inv", lambda f, g: f() ** -1),
])
def test_against_the_reference(key, expr):
  bad = []
  for idx, case in enumerate(CASES):
    got = outcome(lambda: expr(lambda: build(case["f"]), lambda: build(case["g"])))
    if got != case[key]:
      bad.append((idx, case["f"], case["g"], got, case[key]))
  assert not bad, "%d of %d differ; first: %r" % (len(bad), len(CASES), bad[0])


def test_doctest_examples():
  """lazy_filters.py:869-874."""
  from audiolazy_amd import z
  filt = 1 + z ** -1
  assert filt(z ** -1) == z + 1
  assert filt(- z ** 2) == 1 - z ** -2
