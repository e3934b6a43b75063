"""The reference's TestStrategyDict (audiolazy/tests/test_core.py:276-657) restated with
audiolazy_amd.StrategyDict: the design functions (lowpass.pole, resonator["z_exp"], ...) hang on it."""
import operator
from functools import reduce

import pytest

from audiolazy_amd import StrategyDict

p = pytest.mark.parametrize


def test_1x_strategy():                                          # :278-293
  sd = StrategyDict()
  assert len(sd) == 0

  @sd.strategy("test", "t2")
  def sd(a):
    return a + 18
  assert len(sd) == 1
  assert sd["test"](0) == 18 and sd.test(0) == 18 and sd.t2(15) == 33 and sd(-19) == -1
  assert sd.default == sd["test"]


def test_same_key_twice():                                       # :296-320
  sd = StrategyDict()

  @sd.strategy("data", "main", "data")
  def sd():
    return True

  @sd.strategy("only", "only", "main")
  def sd():
    return False
  assert len(sd) == 2
  assert sd["data"] == sd.default and sd["data"] != sd["main"] and sd["only"] == sd["main"]
  assert sd() and sd["data"]() and not sd["only"]() and not sd["main"]()
  assert sd.data() and not sd.only() and not sd.main()
  assert ("data",) in list(sd.keys()) and ("only", "main") in list(sd.keys())


@p("add_names", [("t1", "t2"), ("t1", "t2", "t3")])
@p("mul_names", [("t3",), ("t1", "t2"), ("t1", "t3"), ("t3", "t1"), ("t3", "t2"), ("t1", "t2", "t3"), ("t1")])
def test_2x_strategy(add_names, mul_names):                      # :323-358
  sd = StrategyDict()

  @sd.strategy(*add_names)
  def sd(a, b):
    return a + b

  @sd.strategy(*mul_names)
  def sd(a, b):
    return a * b
  add_names_valid = [name for name in add_names if name not in mul_names]
  assert len(sd) == (1 if len(add_names_valid) == 0 else 2)
  for name in add_names_valid:
    assert sd[name](5, 7) == 12 and sd[name](1, 3) == 4
  for name in mul_names:
    assert sd[name](5, 7) == 35 and sd[name](1, 3) == 3
  if len(add_names_valid) > 0:
    assert sd(-19, 3) == -16
  sd.default = sd[mul_names[0]]
  assert sd(-19, 3) == -57


def test_strategies_names_introspection():                       # :361-394
  sd = StrategyDict()
  sd.strategy("first", "abc")(lambda val: "abc" + val)
  sd.strategy("second", "def")(lambda val: "def" + val)
  sd.strategy("third", "123")(lambda val: "123" + val)
  assert sd("x") == "abcx" and sd.default("p") == "abcp"
  assert sd.first("w") == "abcw" == sd["first"]("w")
  assert sd.second("zsc") == "defzsc" == sd["second"]("zsc")
  assert sd.third("blah") == "123blah" == sd["third"]("blah")
  assert sd.abc("y") == "abcy" == sd["abc"]("y")
  assert sd["def"]("few") == "deffew" and sd["123"]("lots") == "123lots"
  all_names = {"first", "second", "third", "abc", "def", "123"}
  assert all(name in dir(sd) for name in all_names) and all(name in vars(sd) for name in all_names)
  assert "default" in dir(sd) and "default" in vars(sd)
  all_keys_tuples = sd.keys()
  assert set(reduce(operator.concat, all_keys_tuples)) == all_names
  assert set(all_keys_tuples) == {("first", "abc"), ("second", "def"), ("third", "123")}
  assert sd["abc"].__name__ == "first" and sd["def"].__name__ == "second" and sd["123"].__name__ == "third"


def test_empty():                                                # :397-408
  sd = StrategyDict()
  assert "default" in dir(sd) and "default" not in vars(sd)
  assert sd.default() == NotImplemented and sd() == NotImplemented
  assert sd.default(a_key_param="Something") == NotImplemented and sd(some_key_param="Anything") == NotImplemented
  assert sd.default(12) == NotImplemented and sd(34) == NotImplemented
  assert list(sd.keys()) == [] and list(iter(sd)) == []


@p("is_delitem", [True, False])
def test_delitem_delattr(is_delitem):                            # :410-458
  sd = StrategyDict()
  sd.strategy("sum")(lambda *args: reduce(operator.add, args))
  sd.strategy("prod")(lambda *args: reduce(operator.mul, args))
  assert sd.sum(7, 2, 3) == 12 == sd(7, 2, 3) == sd.default(7, 2, 3) and sd.prod(7, 2, 3) == 42
  assert sd["sum"](2, 3) == 5 == sd(2, 3) == sd.default(2, 3) and sd["prod"](2, 3) == 6
  with pytest.raises(KeyError):
    sd["default"](5, 4)
  assert set(sd.keys()) == {("sum",), ("prod",)}
  assert all(n in dir(sd) and n in vars(sd) for n in ("sum", "prod", "default"))
  if is_delitem:
    del sd["sum"]
  else:
    del sd.sum
  assert "sum" not in dir(sd) and "sum" not in vars(sd)
  assert "default" in dir(sd) and "default" not in vars(sd)
  with pytest.raises(AttributeError):
    sd.sum(-1, 2, 3)
  with pytest.raises(KeyError):
    sd["sum"](5, 4)
  with pytest.raises(KeyError):
    sd["default"](5, 4)
  assert list(sd.keys()) == [("prod",)] and len(sd) == 1
  assert "prod" in dir(sd) and "prod" in vars(sd)
  assert sd.prod(-1, 2, 3) == -6 and sd["prod"](5, 4) == 20
  assert sd(3, 2) == NotImplemented == sd.default(3, 2)


def test_strategy_keep_name_and_invalid_kwarg():                 # :460-483
  sd = StrategyDict("sd")
  func = lambda a, b: a + b
  assert func.__name__ == "<lambda>"
  sd.strategy("add", keep_name=True)(func)
  assert func.__name__ == "<lambda>"
  sd.strategy("+", keep_name=False)(func)
  assert func.__name__ == "+" and list(sd.keys()) == [("add", "+")]
  sd.strategy("add")(func)
  assert func.__name__ == "add" and list(sd.keys()) == [("+", "add")]
  sd = StrategyDict("sd")
  identity = lambda x: x
  with pytest.raises(TypeError) as exc:
    sd.strategy("add", weird=True)(identity)
  assert all(w in str(exc.value).lower() for w in ["unknown", "weird"])
  assert len(sd) == 0 and identity.__name__ == "<lambda>"
  assert "default" not in vars(sd) and sd("anything") == NotImplemented


def test_strategy_attribute_replaced():                          # :485-516
  sd = StrategyDict("sd")
  sd.strategy("add", "+", keep_name=True)(operator.add)
  sd.strategy("mul", "*", keep_name=True)(operator.mul)
  sd.strategy("sub", "-", keep_name=True)(operator.sub)
  sd.sub = 14
  assert set(sd.keys()) == {("add", "+"), ("mul", "*"), ("sub", "-",)}
  assert sd["sub"](5, 4) == 1 and sd.sub == 14
  del sd["sub"]
  assert sd.sub == 14 and set(sd.keys()) == {("add", "+"), ("mul", "*"), ("-",)}
  sd.add = None
  assert sd.add is None and sd["add"](3, 7) == 10
  del sd.add
  assert sd.add(4, 7) == 11 == sd["add"](4, 7)
  assert set(sd.keys()) == {("add", "+"), ("mul", "*"), ("-",)}
  del sd.add
  assert set(sd.keys()) == {("+",), ("mul", "*"), ("-",)}
  with pytest.raises(KeyError):
    sd["add"](5, 4)
  with pytest.raises(AttributeError):
    sd.add(5, 4)


def test_non_strategy_delattr():                                 # :518-536
  sd = StrategyDict("sd")
  sd.strategy("add", "+", keep_name=True)(operator.add)
  sd.another = 15
  assert sd.another == 15 and list(sd.keys()) == [("add", "+")]
  with pytest.raises(KeyError):
    del sd["another"]
  sd.another = lambda x: x
  assert sd.another([2, 3, 7]) == [2, 3, 7]
  with pytest.raises(KeyError):
    del sd["another"]
  del sd.another
  assert not hasattr(sd, "another")
  with pytest.raises(AttributeError):
    del sd.another


def test_replacing_default():                                    # :538-575
  sd = StrategyDict("sd")
  sd.strategy("add", "+", keep_name=True)(operator.add)
  sd.strategy("sub", "-", keep_name=True)(operator.sub)
  assert sd(2, 4) == 6
  sd.default = sd.sub
  assert sd(2, 4) == -2
  del sd.sub
  assert sd(3, 4) == -1
  del sd["-"]
  assert sd(7, -3) == NotImplemented
  sd.default = lambda *args: None
  sd.strategy("pow", keep_name=True)(operator.pow)
  assert sd(2, 3) is None
  del sd.default
  assert sd(2, 3) == NotImplemented
  del sd.pow
  sd.strategy("mul", keep_name=True)(operator.mul)
  assert sd(7, -3) == -21
  sd.default = lambda *args, **kwargs: 42
  assert sd(7, -3) == 42
  del sd.mul
  assert len(sd) == 1 and sd(1) == 42
  del sd.add
  del sd["+"]
  assert len(sd) == 0 and sd(3, 2, 1) == 42
  sd.strategy("blah")(sd.default)
  del sd.blah
  assert sd("hua hua hua") == NotImplemented


def test_add_strategy_with_setitem():                            # :577-601
  sdict = StrategyDict("sdict")
  sdict["add"] = operator.add
  sdict["mul"] = operator.mul
  sdict["+"] = operator.add
  assert len(sdict) == 2 and set(sdict.keys()) == {("add", "+"), ("mul",)}
  assert all(name in dir(sdict) and name in vars(sdict) for name in {"add", "+", "mul"})
  assert sdict.add(2, 3) == 5 == sdict["add"](2, 3) and sdict.mul(2, 3) == 6 == sdict["mul"](2, 3)
  assert sdict(7, 8) == 15 == sdict.default(7, 8)
  del sdict["+"]
  assert len(sdict) == 2
  del sdict.add
  assert len(sdict) == 1 and sdict(7, 8) == NotImplemented == sdict.default(7, 8)
  sdict["pow"] = operator.pow
  assert len(sdict) == 2 and sdict(2, 3) == 8 == sdict.default(2, 3)
  assert sdict.pow(5, 2) == 25 == sdict["pow"](5, 2)


@p("use_setitem", [True, False])
def test_reusing_strategy_name(use_setitem):                     # :603-657
  sdict = StrategyDict("sdict")
  m1 = lambda el: el - 1
  p1 = lambda el: el + 1
  if use_setitem:
    sdict["minus_one", "m1"] = m1
    sdict["plus_one", "p1"] = p1
  else:
    sdict.strategy("minus_one", "m1")(m1)
    sdict.strategy("plus_one", "p1")(p1)
  names = {"minus_one", "m1", "plus_one", "p1"}
  assert len(sdict) == 2 and set(sdict.keys()) == {("minus_one", "m1"), ("plus_one", "p1")}
  assert all(name in dir(sdict) and name in vars(sdict) for name in names)
  assert "default" in vars(sdict) and sdict.default == m1
  assert sdict.m1(2) == 1 == sdict["p1"](0) and sdict.p1(2) == 3 == sdict["m1"](4)
  assert sdict(7) == 6 == sdict.default(7)
  if use_setitem:
    sdict["m1"] = p1
  else:
    sdict.strategy("m1")(p1)
  assert len(sdict) == 2 and set(sdict.keys()) == {("minus_one",), ("plus_one", "p1", "m1")}
  assert all(name in dir(sdict) and name in vars(sdict) for name in names)
  assert "default" in vars(sdict) and sdict.default == m1
  assert sdict.m1(2) == 3 == sdict["m1"](2) == sdict["plus_one"](2)
  assert sdict.p1(2) == 3 == sdict["p1"](2) == sdict["minus_one"](4)
  assert sdict(5) == 4 == sdict.default(5)
  if use_setitem:
    sdict["minus_one"] = p1
  else:
    sdict.strategy("minus_one")(p1)
  assert len(sdict) == 1 and list(sdict.keys()) == [("plus_one", "p1", "m1", "minus_one")]
  assert all(name in dir(sdict) and name in vars(sdict) for name in names)
  assert "default" in vars(sdict) and sdict.default == p1
  assert sdict.minus_one(2) == 3 == sdict["m1"](2) == sdict["plus_one"](2)
  assert sdict.plus_one(2) == 3 == sdict["p1"](2) == sdict["minus_one"](2)
  assert sdict(7) == 8 == sdict.default(7)
