"""StrategyDict: named algorithm variants behind one callable.

Mirror of the configuration idiom the reference uses for its design functions
(reference audiolazy/lazy_core.py:431-659): ``lowpass.pole(...)``,
``lowpass["z"](...)``, ``lowpass(...)`` (= the ``default`` strategy) and the
``@name.strategy("a", "alias", ...)`` decorator that registers a function and
rebinds the decorated name to the StrategyDict itself.
"""


class StrategyDict(object):
  def __init__(self, name="strategy_dict"):
    object.__setattr__(self, "_name", name)
    object.__setattr__(self, "_by_name", {})
    object.__setattr__(self, "_default", None)

  def strategy(self, *names):
    def register(func):
      func.__name__ = str(names[0])
      for n in names:
        self._by_name[n] = func
      if self._default is None:
        object.__setattr__(self, "_default", func)
      return self
    return register

  @property
  def default(self):
    return self._default

  @default.setter
  def default(self, func):
    object.__setattr__(self, "_default", func)

  def __call__(self, *args, **kwargs):
    return self._default(*args, **kwargs)

  def __getitem__(self, name):
    return self._by_name[name]

  def __getattr__(self, name):
    try:
      return object.__getattribute__(self, "_by_name")[name]
    except KeyError:
      raise AttributeError("%s has no strategy %r" % (self._name, name))

  def __setattr__(self, name, value):
    if name == "default":
      object.__setattr__(self, "_default", value)
    elif callable(value):
      self._by_name[name] = value
    else:   # plain configuration values, e.g. chunks.size (reference lazy_io.py:45)
      object.__setattr__(self, name, value)

  def __iter__(self):
    seen = []
    for f in self._by_name.values():
      if f not in seen:
        seen.append(f)
    return iter(seen)

  def keys(self):
    return list(self._by_name)

  def __len__(self):
    return len(list(iter(self)))

  def __repr__(self):
    return "<StrategyDict %s: %s>" % (self._name, ", ".join(self._by_name))
