"""``lagrange.func`` / ``lagrange.poly`` / ``resample`` (reference lazy_poly.py:493-603) against tests/golden/lagrange.json, written by
oracle/gen_golden.py --only-lagrange from the unmodified reference: interpolated values with their Python types, the interpolating
Poly's terms, resampled streams bit for bit (constant and time-varying steps)."""
import ast

from conftest import load_golden

G = load_golden("lagrange.json")


def test_interpolating_function_values_and_types():
  from audiolazy_amd import lagrange
  assert lagrange.default is lagrange.func
  for case in G["func"]:
    pairs = [tuple(pr) for pr in ast.literal_eval(case["pairs"])]
    for k, want in zip(ast.literal_eval(case["ks"]), case["values"]):
      try:
        got = repr(lagrange(pairs)(k))
      except Exception as exc:   # noqa: BLE001
        got = "raises " + type(exc).__name__
      assert got == want, (pairs, k)


def test_interpolating_polynomial_terms():
  from audiolazy_amd import lagrange
  for case in G["poly"]:
    pairs = [tuple(pr) for pr in ast.literal_eval(case["pairs"])]
    try:
      got = repr(sorted(lagrange.poly(pairs).terms()))
    except Exception as exc:   # noqa: BLE001
      got = "raises " + type(exc).__name__
    assert got == case["terms"], pairs


def test_resample_bit_for_bit():
  from audiolazy_amd import Stream, resample
  for case in G["resample"]:
    sig = [float.fromhex(v) for v in case["sig"]]
    old = Stream([float.fromhex(v) for v in case["old_series"]]) if "old_series" in case else case["old"]
    got = resample(sig, old=old, new=case["new"], order=case["order"], zero=ast.literal_eval(case["zero"])).take(len(case["y"]))
    assert [float(v).hex() for v in got] == case["y"], {k: case[k] for k in ("new", "order", "zero")}


def test_resample_ends_with_its_input():
  """Twice the rate from six samples: the reference's docstring shape; the stream stops where the input does."""
  from audiolazy_amd import Stream, resample
  out = resample([1, 2, 3, 4, 5, 6], new=2)
  assert isinstance(out, Stream)
  got = list(out)
  assert got[:6] == [1.0, 1.5, 2.0, 2.5, 3.0, 3.5] and len(got) < 40
