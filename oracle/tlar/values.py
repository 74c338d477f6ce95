"""Oracle-R, part 2: TLA+ values.

TEST INFRASTRUCTURE ONLY.  Integers, booleans and strings are Python's; model values, functions
(records and tuples are functions) and the lazily represented sets (Nat, SUBSET S, [S -> T], [a: S, b: T]) are
the classes below.  Every value is immutable and hashable; `sort_key` gives the one enumeration order everything
uses, so runs are deterministic.
"""
from __future__ import annotations

from itertools import product


class TlaEvalError(Exception):
    """What TLC would report as an evaluation error (a function applied outside its domain, CHOOSE over an empty
    set, a non-boolean where a boolean is needed, ...).  The reference's models must never raise it."""


class ModelValue:
    """An (untyped) TLC model value: equal to itself only."""
    __slots__ = ("name",)
    _pool: dict = {}

    def __new__(cls, name):
        mv = cls._pool.get(name)
        if mv is None:
            mv = object.__new__(cls)
            mv.name = name
            cls._pool[name] = mv
        return mv

    def __repr__(self):
        return self.name

    def __reduce__(self):
        return (ModelValue, (self.name,))


class Fn:
    """A function with a finite domain: records have string keys, tuples the keys 1..n."""
    __slots__ = ("d", "_h")

    def __init__(self, d: dict):
        self.d = d
        self._h = None

    def __hash__(self):
        if self._h is None:
            self._h = hash(frozenset(self.d.items()))
        return self._h

    def __eq__(self, other):
        return isinstance(other, Fn) and self.d == other.d

    def __ne__(self, other):
        return not self.__eq__(other)

    def apply(self, k):
        try:
            return self.d[k]
        except KeyError:
            raise TlaEvalError(f"function {self!r} applied to {k!r}, which is outside its domain") from None

    def __repr__(self):
        if self.d and all(isinstance(k, str) for k in self.d):
            return "[" + ", ".join(f"{k} |-> {v!r}" for k, v in self.d.items()) + "]"
        if self.d and sorted(self.d, key=sort_key) == list(range(1, len(self.d) + 1)):
            return "<<" + ", ".join(repr(self.d[i]) for i in range(1, len(self.d) + 1)) + ">>"
        return "(" + " @@ ".join(f"{k!r} :> {v!r}" for k, v in sorted(self.d.items(), key=lambda kv: sort_key(kv[0]))) + ")"


class LazySet:
    """A set that is not enumerated unless it has to be."""

    def contains(self, v):
        raise NotImplementedError

    def elements(self):
        raise TlaEvalError(f"{self!r} cannot be enumerated")

    def __hash__(self):
        return hash(repr(self))

    def __eq__(self, other):
        return type(self) is type(other) and repr(self) == repr(other)


class NatSet(LazySet):
    def contains(self, v):
        return isinstance(v, int) and not isinstance(v, bool) and v >= 0

    def __repr__(self):
        return "Nat"


class IntSet(LazySet):
    def contains(self, v):
        return isinstance(v, int) and not isinstance(v, bool)

    def __repr__(self):
        return "Int"


class PowerSet(LazySet):
    def __init__(self, base):
        self.base = base

    def contains(self, v):
        if isinstance(v, frozenset):
            return all(set_contains(self.base, x) for x in v)
        return False

    def elements(self):
        xs = set_elements(self.base)
        out = []
        for mask in range(1 << len(xs)):
            out.append(frozenset(x for i, x in enumerate(xs) if mask >> i & 1))
        return out

    def __repr__(self):
        return f"SUBSET {self.base!r}"


class FuncSet(LazySet):
    def __init__(self, dom, rng):
        self.dom, self.rng = dom, rng

    def contains(self, v):
        if not isinstance(v, Fn):
            return False
        dom = set_elements(self.dom)
        if len(v.d) != len(dom) or any(k not in v.d for k in dom):
            return False
        return all(set_contains(self.rng, x) for x in v.d.values())

    def elements(self):
        dom = set_elements(self.dom)
        rng = set_elements(self.rng)
        return [Fn(dict(zip(dom, vals))) for vals in product(rng, repeat=len(dom))]

    def __repr__(self):
        return f"[{self.dom!r} -> {self.rng!r}]"


class RecordSet(LazySet):
    def __init__(self, fields):
        self.fields = fields  # [(name, set)]

    def contains(self, v):
        if not isinstance(v, Fn) or len(v.d) != len(self.fields):
            return False
        return all(k in v.d and set_contains(s, v.d[k]) for k, s in self.fields)

    def elements(self):
        names = [k for k, _ in self.fields]
        return [Fn(dict(zip(names, vals))) for vals in product(*(set_elements(s) for _, s in self.fields))]

    def __repr__(self):
        return "[" + ", ".join(f"{k}: {s!r}" for k, s in self.fields) + "]"


# ------------------------------------------------------------------------------------------------
def is_set(v):
    return isinstance(v, (frozenset, LazySet))


def set_contains(s, v):
    if isinstance(s, frozenset):
        return v in s
    if isinstance(s, LazySet):
        return s.contains(v)
    raise TlaEvalError(f"{s!r} is not a set")


_sorted_cache: dict = {}


def set_elements(s):
    """The elements of a set in the canonical order (a list; do not mutate)."""
    if isinstance(s, frozenset):
        got = _sorted_cache.get(s)
        if got is None:
            got = sorted(s, key=sort_key)
            if len(_sorted_cache) > 200000:
                _sorted_cache.clear()
            _sorted_cache[s] = got
        return got
    if isinstance(s, LazySet):
        return sorted(s.elements(), key=sort_key)
    raise TlaEvalError(f"{s!r} is not a set")


def as_frozenset(s):
    if isinstance(s, frozenset):
        return s
    return frozenset(set_elements(s))


def sort_key(v):
    if isinstance(v, bool):
        return (0, int(v))
    if isinstance(v, int):
        return (1, v)
    if isinstance(v, str):
        return (2, v)
    if isinstance(v, ModelValue):
        return (3, v.name)
    if isinstance(v, frozenset):
        return (4, len(v), tuple(sort_key(x) for x in set_elements(v)))
    if isinstance(v, Fn):
        items = sorted(((sort_key(k), sort_key(x)) for k, x in v.d.items()))
        return (5, len(items), tuple(items))
    return (6, repr(v))


def values_equal(a, b):
    """TLA+ equality.  Sets compare extensionally whatever their representation; a boolean never equals an integer."""
    if isinstance(a, LazySet) or isinstance(b, LazySet):
        if not (is_set(a) and is_set(b)):
            return False
        return as_frozenset(a) == as_frozenset(b)
    if isinstance(a, bool) != isinstance(b, bool):
        return False
    return a == b


def fmt(v):
    """TLC-ish rendering (for traces and error messages)."""
    if isinstance(v, bool):
        return "TRUE" if v else "FALSE"
    if isinstance(v, str):
        return '"' + v + '"'
    if isinstance(v, frozenset):
        return "{" + ", ".join(fmt(x) for x in set_elements(v)) + "}"
    if isinstance(v, Fn):
        d = v.d
        if d and all(isinstance(k, str) for k in d):
            return "[" + ", ".join(f"{k} |-> {fmt(x)}" for k, x in d.items()) + "]"
        if d and set(d) == set(range(1, len(d) + 1)):
            return "<<" + ", ".join(fmt(d[i]) for i in range(1, len(d) + 1)) + ">>"
        return "(" + " @@ ".join(f"{fmt(k)} :> {fmt(x)}" for k, x in sorted(d.items(), key=lambda kv: sort_key(kv[0]))) + ")"
    return repr(v)
