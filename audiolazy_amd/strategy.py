"""StrategyDict: named algorithm variants behind one callable.

Mirror of the configuration idiom the reference uses for its design functions
(reference audiolazy/lazy_core.py:431-659): ``lowpass.pole(...)``, ``lowpass["z"](...)``,
``lowpass(...)`` (= the ``default`` strategy) and the ``@name.strategy("a", "alias", ...)``
decorator that registers a function and rebinds the decorated name to the StrategyDict itself.

The bookkeeping follows the reference's: a strategy is a function known under one or more names
(``keys()`` yields one tuple of names per strategy, in insertion order); giving a name to another
function takes it away from the previous one, and a function left without names is gone;
registering an already known function under a new name extends its tuple.  The names are also
instance attributes (so ``vars()`` / ``dir()`` list them); an attribute can be overwritten with a
plain value without touching the strategy behind it, and deleted again.  ``default`` is the first
strategy registered unless set explicitly; without one, calling gives ``NotImplemented``.
"""


def _not_implemented(*args, **kwargs):
  return NotImplemented


class StrategyDict(object):
  def __init__(self, name="strategy_dict_unnamed_instance"):
    self.__dict__["_sd_name"] = name
    self.__dict__["_sd_entries"] = []     # [[names...], function] in insertion order

  # -- registry ------------------------------------------------------------------
  def _find(self, name):
    for entry in self._sd_entries:
      if name in entry[0]:
        return entry
    return None

  def _drop_name(self, name, successor=None):
    """Take ``name`` away from its strategy; a strategy without names disappears (and with it
    the default, if it was the default: ``successor`` -- the function taking over the name --
    inherits it, a plain deletion leaves no default)."""
    entry = self._find(name)
    if entry is None:
      return
    entry[0].remove(name)
    if self.__dict__.get(name) is entry[1]:
      del self.__dict__[name]
    if not entry[0]:
      self._sd_entries[:] = [e for e in self._sd_entries if e is not entry]
      if self.__dict__.get("default") is entry[1]:
        if successor is not None:
          self.__dict__["default"] = successor
        else:
          del self.__dict__["default"]

  def __setitem__(self, names, func):
    if isinstance(names, str):
      names = (names,)
    for name in names:
      self._drop_name(name, successor=func)
    entry = None
    for candidate in self._sd_entries:
      if candidate[1] is func:
        entry = candidate
    if entry is None:
      entry = [[], func]
      self._sd_entries.append(entry)
    for name in names:
      if name not in entry[0]:
        entry[0].append(name)
      self.__dict__[name] = func
    if "default" not in self.__dict__:
      self.__dict__["default"] = func

  def strategy(self, *names, **kwargs):
    keep_name = kwargs.pop("keep_name", False)
    if kwargs:
      raise TypeError("Unknown keyword argument '%s'" % sorted(kwargs)[0])

    def register(func):
      if not keep_name:
        func.__name__ = str(names[0])
      self[names] = func
      return self
    return register

  def __getitem__(self, name):
    entry = self._find(name)
    if entry is None:
      raise KeyError(name)
    return entry[1]

  def __delitem__(self, name):
    if self._find(name) is None:
      raise KeyError(name)
    self._drop_name(name)

  # -- attributes ----------------------------------------------------------------
  @property
  def default(self):
    return self.__dict__.get("default", _not_implemented)

  @default.setter
  def default(self, func):
    self.__dict__["default"] = func

  @default.deleter
  def default(self):
    self.__dict__.pop("default", None)

  def __getattr__(self, name):     # only reached when the instance has no such attribute
    raise AttributeError("%s has no strategy %r" % (self.__dict__.get("_sd_name", "StrategyDict"), name))

  def __setattr__(self, name, value):
    if name == "default":
      self.__dict__["default"] = value
    else:                          # a plain attribute: strategies are registered with strategy() / []
      self.__dict__[name] = value

  def __delattr__(self, name):
    if name == "default":
      self.__dict__.pop("default", None)
      return
    entry = self._find(name)
    if name in self.__dict__ and (entry is None or self.__dict__[name] is not entry[1]):
      if entry is not None:
        self.__dict__[name] = entry[1]    # the overwritten attribute goes, the strategy shows again
      else:
        del self.__dict__[name]
      return
    if entry is None:
      raise AttributeError(name)
    self._drop_name(name)

  # -- behaviour -----------------------------------------------------------------
  def __call__(self, *args, **kwargs):
    return self.default(*args, **kwargs)

  def __iter__(self):
    return iter([entry[1] for entry in self._sd_entries])

  def keys(self):
    return [tuple(entry[0]) for entry in self._sd_entries]

  def __len__(self):
    return len(self._sd_entries)

  def __repr__(self):
    return "<StrategyDict %s: %s>" % (self._sd_name, ", ".join("/".join(k) for k in self.keys()))
