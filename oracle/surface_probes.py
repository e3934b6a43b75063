"""Probes of the Poly / filter-object attribute surface (TEST INFRASTRUCTURE: imported by oracle/gen_golden.py, which
runs them on the reference and commits the outcomes as tests/golden/poly_surface.json / filter_surface.json, by
tests/test_surface_golden.py, which runs them on audiolazy_amd, and by tools/fuzz_poly.py / fuzz_surface.py, which run
both side by side on fresh random cases in the build container).  Every probe's outcome is a JSON-able value with the
Python type of every number, or ("raises", exception name).  Reference lines: audiolazy/lazy_poly.py:66-490,
lazy_filters.py:47-95, 110-338, 692-892, 895-1084."""
import random


def to_json(v):
  """tuples -> lists, recursively (what a JSON round trip gives)."""
  if isinstance(v, (list, tuple)):
    return [to_json(x) for x in v]
  return v


def outcome(norm, mod, fn):
  try:
    return to_json(norm(mod, fn()))
  except Exception as exc:   # noqa: BLE001 -- the exception type IS the datum
    return ["raises", type(exc).__name__]


# ---------------------------------------------------------------- Poly ----------------------------------------------
def poly_norm(mod, v):
  if isinstance(v, bool) or v is None:
    return v
  if isinstance(v, int):
    return ("i", v)
  if isinstance(v, float):
    return ("f", v.hex())
  if isinstance(v, complex):
    return ("c", v.real.hex(), v.imag.hex())
  if isinstance(v, mod.Poly):
    return ("P", [(poly_norm(mod, k), poly_norm(mod, c)) for k, c in v.terms()], poly_norm(mod, getattr(v, "zero", "nozero")))
  if isinstance(v, dict):
    return ("d", [(poly_norm(mod, k), poly_norm(mod, c)) for k, c in v.items()])
  if isinstance(v, (list, tuple)):
    return ("l", [poly_norm(mod, x) for x in v])
  if hasattr(v, "__next__"):
    return ("g", [poly_norm(mod, x) for x in v])
  if isinstance(v, str):
    return ("s", v)
  return ("?", type(v).__name__)


def poly_case(rng):
  integer = rng.random() < .4
  def coef():
    if integer:
      return rng.choice([-3, -2, -1, 0, 1, 2, 3])
    return rng.choice([rng.uniform(-2., 2.), round(rng.uniform(-3., 3.), 1), .5, -.25, 0., 1.5])
  kind = rng.choice(["list", "dict", "dict", "laurent", "number", "empty", "frac"])
  if kind == "list":
    return [coef() for _ in range(rng.randint(0, 4))]
  if kind == "dict":
    return {p: coef() for p in rng.sample(range(0, 5), rng.randint(0, 3))}
  if kind == "laurent":
    return {p: coef() for p in rng.sample(range(-3, 4), rng.randint(1, 3))}
  if kind == "frac":
    return {rng.choice([.5, 1.5, -0.5, 2.0, 3.0, 0.0]): coef(), 1: coef()}
  if kind == "number":
    return coef()
  return None


POLY_PROBES = {
  "terms": lambda m, p, q: list(p.terms()), "terms_rev": lambda m, p, q: list(p.terms(reverse=True)) if "reverse" else None,
  "values": lambda m, p, q: list(p.values()), "order": lambda m, p, q: p.order,
  "is_polynomial": lambda m, p, q: p.is_polynomial(), "is_laurent": lambda m, p, q: p.is_laurent(),
  "len": lambda m, p, q: len(p), "zero": lambda m, p, q: p.zero,
  "add": lambda m, p, q: p + q, "sub": lambda m, p, q: p - q, "mul": lambda m, p, q: p * q,
  "neg": lambda m, p, q: -p, "pos": lambda m, p, q: +p,
  "add_n": lambda m, p, q: p + 2, "radd_n": lambda m, p, q: 2.5 + p, "rsub_n": lambda m, p, q: 1 - p, "sub_n": lambda m, p, q: p - 1.5,
  "mul_n": lambda m, p, q: p * 3, "rmul_n": lambda m, p, q: .5 * p, "div_n": lambda m, p, q: p / 2, "div_nf": lambda m, p, q: p / .3,
  "div_p": lambda m, p, q: p / q, "rdiv": lambda m, p, q: 2 / p, "div0": lambda m, p, q: p / 0,
  "pow2": lambda m, p, q: p ** 2, "pow3": lambda m, p, q: p ** 3, "pow0": lambda m, p, q: p ** 0, "pow1": lambda m, p, q: p ** 1,
  "powm1": lambda m, p, q: p ** -1, "powh": lambda m, p, q: p ** .5, "rpow": lambda m, p, q: 2 ** p,
  "call_i": lambda m, p, q: p(2), "call_f": lambda m, p, q: p(-.75), "call_0": lambda m, p, q: p(0), "call_c": lambda m, p, q: p(1j),
  "call_p": lambda m, p, q: p(q), "call_x": lambda m, p, q: p(m.x + 1), "call_nohorner": lambda m, p, q: p(1.5, horner=False),
  "call_horner": lambda m, p, q: p(1.5, horner=True),
  "diff": lambda m, p, q: p.diff(), "diff2": lambda m, p, q: p.diff(2), "diff_x": lambda m, p, q: p.diff(n=1, ),
  "integrate": lambda m, p, q: p.integrate(),
  "copy": lambda m, p, q: p.copy(), "copy_zero": lambda m, p, q: p.copy(zero=0),
  "eq": lambda m, p, q: p == q, "ne": lambda m, p, q: p != q, "eq_self": lambda m, p, q: p == p.copy(), "eq_num": lambda m, p, q: p == 1, "ne_num": lambda m, p, q: p != 0,
  "getitem": lambda m, p, q: [p[k] for k in (-1, 0, 1, 2, 7)],
  "roots": lambda m, p, q: sorted((round(r.real, 9), round(r.imag, 9)) for r in map(complex, p.roots)),
  "x_expr": lambda m, p, q: (m.x ** 2 - 3 * m.x + 1.5) * p,
  "hash_eq": lambda m, p, q: hash(p) == hash(p.copy()),
  "zero_arg": lambda m, p, q: m.Poly(list(p.values()), zero=0),
  "setitem": lambda m, p, q: (lambda c: (c.__setitem__(2, 7), c)[1])(p.copy()),
  "setitem_zero": lambda m, p, q: (lambda c: (c.__setitem__(1, 0.), c)[1])(p.copy()),
  "str": lambda m, p, q: str(p),
}




def poly_outcomes(mod, seed, n):
  rng = random.Random(seed)
  rows = []
  for _ in range(n):
    a, b = poly_case(rng), poly_case(rng)
    rows.append({name: outcome(poly_norm, mod, lambda: probe(mod, mod.Poly(a), mod.Poly(b))) for name, probe in POLY_PROBES.items()})
  return rows


# ------------------------------------------------------- ZFilter / CascadeFilter / ParallelFilter -------------------
def filter_norm(mod, v, depth=0):
  """A comparable, module-independent form of whatever an attribute returned."""
  if isinstance(v, bool) or v is None:
    return v
  if isinstance(v, int):
    return ("i", v)
  if isinstance(v, float):
    return ("f", v.hex())
  if isinstance(v, complex):
    return ("c", v.real.hex(), v.imag.hex())
  if isinstance(v, mod.ZFilter):
    return ("Z", [(filter_norm(mod, k), filter_norm(mod, c)) for k, c in v.numpoly.terms()],
            [(filter_norm(mod, k), filter_norm(mod, c)) for k, c in v.denpoly.terms()])
  if isinstance(v, (mod.CascadeFilter, mod.ParallelFilter)):
    return (type(v).__name__, [filter_norm(mod, f, depth + 1) for f in v])
  if isinstance(v, mod.Poly):
    return ("P", [(filter_norm(mod, k), filter_norm(mod, c)) for k, c in v.terms()])
  if isinstance(v, dict):
    return ("d", sorted((filter_norm(mod, k), filter_norm(mod, c)) for k, c in v.items()))
  if isinstance(v, str):
    return ("s", v)
  if isinstance(v, (list, tuple)):
    return ("l", [filter_norm(mod, x, depth + 1) for x in v])
  if hasattr(v, "take") and hasattr(v, "__next__") or type(v).__name__ in ("Stream", "ControlStream"):
    return ("S", [filter_norm(mod, x) for x in v.take(5)])
  return ("?", type(v).__name__)




def random_rational(rng, kind):
  """(num terms, den terms) as {power: value} dicts in z ** -1 (the generator of composition.json, oracle/gen_golden.py)."""
  def coef(integer):
    if integer:
      return rng.choice([-3, -2, -1, 1, 2, 3, 4])
    return rng.choice([rng.uniform(-2., 2.), round(rng.uniform(-3., 3.), 1), .5, -.25, 1.5])
  def poly(nterms, integer, lo, hi):
    return {p: coef(integer) for p in rng.sample(range(lo, hi + 1), nterms)}
  integer = rng.random() < .4
  if kind == "const":
    return {0: coef(integer)}, {0: 1}
  if kind == "delay":
    return {rng.choice([-3, -2, -1, 1, 2, 3]): rng.choice([1, 1, -1, coef(integer)])}, {0: 1}
  if kind == "zero":
    return {}, {0: 1} if rng.random() < .5 else poly(2, integer, 0, 2)
  if kind == "fir":
    return poly(rng.randint(1, 3), integer, -1, 3), {0: 1}
  num = poly(rng.randint(1, 3), integer, -1, 3)
  den = poly(rng.randint(1, 3), integer, 0, 3)
  if rng.random() < .3:
    den[0] = 1
  return num, den


def filter_build(mod, spec):
  kind, parts = spec
  fs = [mod.ZFilter(dict(n), dict(d)) for n, d in parts]
  if kind == "z":
    return fs[0]
  return (mod.CascadeFilter if kind == "cascade" else mod.ParallelFilter)(fs)


FILTER_PROBES = {
  "numlist": lambda m, f: f.numlist, "denlist": lambda m, f: f.denlist,
  "numdict": lambda m, f: f.numdict, "dendict": lambda m, f: f.dendict,
  "numpoly": lambda m, f: f.numpoly, "denpoly": lambda m, f: f.denpoly,
  "numpolyz": lambda m, f: f.numpolyz, "denpolyz": lambda m, f: f.denpolyz,
  "numerator": lambda m, f: f.numerator, "denominator": lambda m, f: f.denominator,
  "is_causal": lambda m, f: f.is_causal(), "is_lti": lambda m, f: f.is_lti(),
  "is_linear": lambda m, f: f.is_linear() if hasattr(f, "is_linear") else None,
  "freq_response": lambda m, f: [f.freq_response(w) for w in (0., .3, 1.7, 3.141592653589793)],
  "freq_response_list": lambda m, f: list(f.freq_response([.1, .2])) if True else None,
  "diff": lambda m, f: f.diff(), "diff2": lambda m, f: f.diff(n=2, mul_after=-m.z),
  "linearize": lambda m, f: f.linearize(), "copy": lambda m, f: f.copy(),
  "neg": lambda m, f: -f, "times2": lambda m, f: f * 2, "rtimes": lambda m, f: 2.5 * f,
  "plus1": lambda m, f: f + 1, "rminus": lambda m, f: 1 - f, "div3": lambda m, f: f / 3, "rdiv": lambda m, f: 3 / f,
  "pow2": lambda m, f: f ** 2, "powm1": lambda m, f: f ** -1, "pow0": lambda m, f: f ** 0,
  "times_z": lambda m, f: f * m.z ** -1, "plus_z": lambda m, f: f + m.z ** -2,
  "len": lambda m, f: len(f) if hasattr(f, "__len__") else None,
  "str": lambda m, f: str(f) if isinstance(f, m.ZFilter) else None, "repr": lambda m, f: repr(f) if isinstance(f, m.ZFilter) else None,
  "poles": lambda m, f: f.poles, "zeros": lambda m, f: f.zeros,
  "ne_self": lambda m, f: f != f.copy(), "ne_num": lambda m, f: f != f * 2, "ne_other": lambda m, f: f != 5, "eq_other": lambda m, f: f == 5,
  "ne_both": lambda m, f: f != (f * 2) / (1 + m.z ** -1),
  "eq_self": lambda m, f: f == f.copy() if hasattr(f, "copy") else None,
}




def filter_case(rng):
  kinds = ["const", "delay", "zero", "fir", "rational", "rational", "rational", "fir"]
  kind = rng.choice(["z", "z", "cascade", "parallel"])
  parts = [random_rational(rng, rng.choice(kinds)) for _ in range(1 if kind == "z" else rng.randint(0, 3))]
  return (kind, parts)


def filter_outcomes(mod, seed, n):
  rng = random.Random(seed)
  rows = []
  for _ in range(n):
    spec = filter_case(rng)
    rows.append({name: outcome(filter_norm, mod, lambda: probe(mod, filter_build(mod, spec))) for name, probe in FILTER_PROBES.items()})
  return rows
