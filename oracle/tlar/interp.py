"""Oracle-R, part 3: module loading (EXTENDS / INSTANCE ... WITH / LOCAL) and the evaluator.

TEST INFRASTRUCTURE ONLY.  Evaluates the reference's own text the way TLC's interpreter does [TLC-recall]:

  * a state predicate / value expression is evaluated against a state (`ev`);
  * an action is evaluated into the stream of its successor states (`act`): conjuncts left to right, every disjunct,
    every binding of a bounded \\E, `x' = e` (and UNCHANGED) on a not-yet-assigned variable assigns it, anything else
    is a condition on what is already assigned.  One successor per satisfying binding — duplicates included, which is
    what TLC's "states generated" counts;
  * `Init` is evaluated by the same machinery with unprimed variables in the role of the primed ones.

Operator arguments and LET definitions are call-by-need (TLC evaluates them lazily too), so a definition such as
`LET offset == LookupOffsetForEpoch(...) IN BecomeFollowerAndTruncateTo(leader, replica, offset)` is only evaluated
behind the guards that protect it.
"""
from __future__ import annotations

import os

from .syntax import Def, InstanceDef, ModuleAst, Node, parse_module
from .values import (Fn, FuncSet, IntSet, LazySet, ModelValue, NatSet, PowerSet, RecordSet, TlaEvalError,
                     as_frozenset, is_set, set_contains, set_elements, sort_key, values_equal)

STANDARD_MODULES = {"Integers", "Naturals", "FiniteSets", "Sequences", "TLC"}  # only their arithmetic is used


class Module:
    """A loaded module: its own definitions plus everything it imports by EXTENDS."""

    def __init__(self, ast: ModuleAst):
        self.name = ast.name
        self.ast = ast
        self.own = {}        # name -> Def | InstanceDef (including LOCAL ones)
        self.exported = {}   # what an extender / instantiator sees: non-LOCAL own + exported of the extended modules
        self.visible = {}    # what this module's own expressions see: own + exported of the extended modules
        self.constants = []  # declared here or in an extended module
        self.variables = []

    def __repr__(self):
        return f"Module({self.name})"


class Loader:
    def __init__(self, search_path):
        self.search_path = list(search_path)
        self.modules = {}

    def load(self, name) -> Module:
        if name in self.modules:
            return self.modules[name]
        for d in self.search_path:
            fn = os.path.join(d, name + ".tla")
            if os.path.exists(fn):
                break
        else:
            raise FileNotFoundError(f"module {name}.tla not found in {self.search_path}")
        with open(fn) as f:
            ast = parse_module(f.read(), fn)
        if ast.name != name:
            raise TlaEvalError(f"{fn} declares module {ast.name}")
        m = Module(ast)
        self.modules[name] = m
        for ext in ast.extends:
            if ext in STANDARD_MODULES:
                continue
            e = self.load(ext)
            for k, d in e.exported.items():
                m.exported[k] = d
                m.visible[k] = d
            for c in e.constants:
                if c not in m.constants:
                    m.constants.append(c)
            for v in e.variables:
                if v not in m.variables:
                    m.variables.append(v)
        m.constants += [c for c in ast.constants if c not in m.constants]
        m.variables += [v for v in ast.variables if v not in m.variables]
        for d in ast.defs:
            d.home = m
            if d.name in m.own:
                raise TlaEvalError(f"{fn}: {d.name} defined twice")
            m.own[d.name] = d
            m.visible[d.name] = d
            if not d.local:
                m.exported[d.name] = d
            if isinstance(d, InstanceDef):
                self.load(d.target)
        return m


class Frame:
    """One INSTANCE: the instantiated module's constants and variables are replaced by expressions of the
    instantiating module (explicit WITH, else the same name), evaluated in `outer`."""
    __slots__ = ("inst", "outer_module", "outer_frame", "key")

    def __init__(self, inst: InstanceDef, outer_module: Module, outer_frame):
        self.inst, self.outer_module, self.outer_frame = inst, outer_module, outer_frame
        self.key = (id(inst), outer_frame.key if outer_frame else None)


class Thunk:
    """A call-by-need argument or LET definition without parameters."""
    __slots__ = ("expr", "module", "frame", "env", "val", "done")

    def __init__(self, expr, module, frame, env):
        self.expr, self.module, self.frame, self.env = expr, module, frame, env
        self.val, self.done = None, False


class LetOp:
    """A LET definition with parameters."""
    __slots__ = ("d", "module", "frame", "env")

    def __init__(self, d, module, frame, env):
        self.d, self.module, self.frame, self.env = d, module, frame, env


_NAT, _INT = NatSet(), IntSet()
_BOOLEAN = frozenset((False, True))


class Interp:
    def __init__(self, loader: Loader, root: str, constants: dict):
        self.loader = loader
        self.root = loader.load(root)
        self.constants = dict(constants)
        missing = [c for c in self.root.constants if c not in self.constants]
        if missing:
            raise TlaEvalError(f"no value for CONSTANT(S) {missing} of {root}")
        self.variables = list(self.root.variables)
        self.cur = None         # the current state: dict var -> value
        self.init_mode = False  # evaluating Init: unprimed variables are the ones being assigned
        self.state_reads = 0    # bumped on every variable read (constant-definition cache)
        self.const_cache = {}
        self.frames = {}

    # ---- public API -----------------------------------------------------------------------
    def check_assumes(self):
        """ASSUME of the root module and of everything it extends (not of instantiated modules: their constants are
        substituted, and TLC checks the instantiated assumptions only through the root's theorems)."""
        bad = []
        seen = set()

        def walk(m):
            if m.name in seen:
                return
            seen.add(m.name)
            for e in m.ast.extends:
                if e not in STANDARD_MODULES:
                    walk(self.loader.load(e))
            for a in m.ast.assumes:
                if self.ev(a, m, None, {}, None, False) is not True:
                    bad.append((m.name, a.line))
        walk(self.root)
        return bad

    def initial_states(self, init="Init"):
        self.init_mode, self.cur = True, None
        try:
            out = []
            for nxt in self.act(Node("ident", init), self.root, None, {}, {}, None):
                self._complete(nxt, init)
                out.append(dict(nxt))
            return out
        finally:
            self.init_mode = False

    def successors(self, state: dict, next_="Next"):
        """[(label, successor dict)]; label = the top-level disjunct of Next that produced it (its operator name when it
        is one), None when Next has no disjunction."""
        self.cur = state
        out = []
        label = [None]
        for nxt in self.act(Node("ident", next_), self.root, None, {}, {}, label):
            self._complete(nxt, next_)
            out.append((label[0], nxt))
        return out

    def holds(self, state: dict, name: str):
        self.cur = state
        v = self.ev(Node("ident", name), self.root, None, {}, None, False)
        if not isinstance(v, bool):
            raise TlaEvalError(f"{name} is not boolean: {v!r}")
        return v

    def _complete(self, nxt, what):
        for v in self.variables:
            if v not in nxt:
                raise TlaEvalError(f"{what} leaves variable {v} unassigned")

    # ---- name resolution ------------------------------------------------------------------
    def frame_for(self, inst: InstanceDef, frame):
        key = (id(inst), frame.key if frame else None)
        f = self.frames.get(key)
        if f is None:
            f = self.frames[key] = Frame(inst, inst.home, frame)
        return f

    def resolve_var(self, e: Node, module: Module, frame, env):
        """The root variable an expression denotes after INSTANCE substitution (None when it is not a bare variable)."""
        while e.kind == "paren":
            e = e.a
        if e.kind != "ident" or e.a in env:
            return None
        name = e.a
        if name in module.visible:
            return None
        if name in module.variables or name in module.constants:
            if frame is None:
                return name if name in self.variables else None
            sub = frame.inst.substs.get(name) or Node("ident", name)
            return self.resolve_var(sub, frame.outer_module, frame.outer_frame, {})
        return None

    def lookup(self, name, module, frame, env, nxt, primed, node):
        """Value of a bare identifier."""
        if name in env:
            b = env[name]
            if isinstance(b, Thunk):
                return self.force(b, nxt, primed)
            if isinstance(b, LetOp):
                raise TlaEvalError(f"{name} needs arguments")
            return b
        d = module.visible.get(name)
        if d is not None:
            if isinstance(d, InstanceDef):
                raise TlaEvalError(f"instance {name} used as a value")
            if d.params:
                raise TlaEvalError(f"{name} needs {len(d.params)} argument(s)")
            return self.eval_def0(d, frame, nxt, primed)
        if name in module.variables or name in module.constants:
            if frame is not None:
                sub = frame.inst.substs.get(name) or Node("ident", name)
                return self.ev(sub, frame.outer_module, frame.outer_frame, {}, nxt, primed)
            if name in self.constants:
                return self.constants[name]
            self.state_reads += 1
            src = nxt if (primed or self.init_mode) else self.cur
            if src is None or name not in src:
                raise TlaEvalError(f"variable {name}{chr(39) if primed else ''} read before it is assigned "
                                   f"(line {node.line})")
            return src[name]
        if name == "Nat":
            return _NAT
        if name == "Int":
            return _INT
        if name == "BOOLEAN":
            return _BOOLEAN
        raise TlaEvalError(f"unknown identifier {name} in module {module.name} (line {node.line})")

    def eval_def0(self, d: Def, frame, nxt, primed):
        key = (id(d), frame.key if frame else None)
        got = self.const_cache.get(key, self)
        if got is not self:
            return got
        before = self.state_reads
        v = self.ev(d.body, d.home, frame, {}, nxt, primed)
        if self.state_reads == before:
            self.const_cache[key] = v
        return v

    def force(self, t: Thunk, nxt, primed):
        if primed:
            return self.ev(t.expr, t.module, t.frame, t.env, nxt, True)
        if not t.done:
            t.val = self.ev(t.expr, t.module, t.frame, t.env, nxt, False)
            t.done = True
        return t.val

    def find_operator(self, node, module, frame, env):
        """-> (def, home module, frame, base env, param names) for apply / inst / ident nodes naming an operator with a
        body; None for built-ins and plain values."""
        if node.kind == "inst":
            m, f = module, frame
            for iname in node.a:
                inst = m.visible.get(iname)
                if not isinstance(inst, InstanceDef):
                    raise TlaEvalError(f"{iname} is not an instance (line {node.line})")
                f = self.frame_for(inst, f)
                m = self.loader.load(inst.target)
                vis = m.exported
            d = m.exported.get(node.b)
            if d is None:
                raise TlaEvalError(f"{'!'.join(node.a)}!{node.b} is not defined (line {node.line})")
            return d, d.home, f, {}, d.params
        name = node.a
        if name in env:
            b = env[name]
            if isinstance(b, LetOp):
                return b.d, b.module, b.frame, b.env, b.d.params
            return None
        d = module.visible.get(name)
        if isinstance(d, Def):
            return d, d.home, frame, {}, d.params
        return None

    def bind_args(self, node, params, args, base_env, module, frame, env):
        if len(params) != len(args):
            raise TlaEvalError(f"operator at line {node.line} takes {len(params)} argument(s), given {len(args)}")
        if not params:
            return base_env
        new = dict(base_env)
        for p, a in zip(params, args):
            new[p] = Thunk(a, module, frame, env)
        return new

    # ---- values ---------------------------------------------------------------------------
    def ev_bool(self, e, module, frame, env, nxt, primed):
        v = self.ev(e, module, frame, env, nxt, primed)
        if v is True or v is False:
            return v
        raise TlaEvalError(f"expected a boolean at line {e.line}, got {v!r}")

    def ev(self, e: Node, module, frame, env, nxt, primed):
        k = e.kind
        if k == "ident":
            return self.lookup(e.a, module, frame, env, nxt, primed, e)
        if k == "num" or k == "str" or k == "bool":
            return e.a
        if k == "paren":
            return self.ev(e.a, module, frame, env, nxt, primed)
        if k == "op":
            return self.ev_op(e, module, frame, env, nxt, primed)
        if k == "and":
            for it in e.a:
                if not self.ev_bool(it, module, frame, env, nxt, primed):
                    return False
            return True
        if k == "or":
            for it in e.a:
                if self.ev_bool(it, module, frame, env, nxt, primed):
                    return True
            return False
        if k == "dot":
            r = self.ev(e.a, module, frame, env, nxt, primed)
            if not isinstance(r, Fn):
                raise TlaEvalError(f"field {e.b} of a non-record {r!r} (line {e.line})")
            return r.apply(e.b)
        if k == "fapp":
            f = self.ev(e.a, module, frame, env, nxt, primed)
            args = [self.ev(x, module, frame, env, nxt, primed) for x in e.b]
            if not isinstance(f, Fn):
                raise TlaEvalError(f"{f!r} is not a function (line {e.line})")
            return f.apply(args[0] if len(args) == 1 else Fn({i + 1: a for i, a in enumerate(args)}))
        if k == "apply" or k == "inst":
            op = self.find_operator(e, module, frame, env)
            args = e.c if k == "inst" else e.b
            if op is None:
                raise TlaEvalError(f"unknown operator {e.a} (line {e.line})")
            d, home, f2, base, params = op
            if k == "inst" and not params and not args and isinstance(d, Def):
                return self.eval_def0(d, f2, nxt, primed)
            new_env = self.bind_args(e, params, args, base, module, frame, env)
            return self.ev(d.body, home, f2, new_env, nxt, primed)
        if k == "prime":
            if nxt is None:
                raise TlaEvalError(f"primed expression in a state-level context (line {e.line})")
            return self.ev(e.a, module, frame, env, nxt, True)
        if k == "if":
            c = self.ev_bool(e.a, module, frame, env, nxt, primed)
            return self.ev(e.b if c else e.c, module, frame, env, nxt, primed)
        if k == "let":
            return self.ev(e.b, module, frame, self.let_env(e, module, frame, env), nxt, primed)
        if k == "quant":
            return self.ev_quant(e, 0, 0, module, frame, env, nxt, primed)
        if k == "choose":
            s = self.ev(e.b, module, frame, env, nxt, primed)
            for x in set_elements(s):
                if self.ev_bool(e.c, module, frame, {**env, e.a: x}, nxt, primed):
                    return x
            raise TlaEvalError(f"CHOOSE over a set with no witness (line {e.line})")
        if k == "setenum":
            return frozenset(self.ev(x, module, frame, env, nxt, primed) for x in e.a)
        if k == "setfilter":
            s = self.ev(e.b, module, frame, env, nxt, primed)
            return frozenset(x for x in set_elements(s) if self.ev_bool(e.c, module, frame, {**env, e.a: x}, nxt, primed))
        if k == "setmap":
            out = []
            self.each_binding(e.b, 0, 0, module, frame, env, nxt, primed,
                              lambda env2: out.append(self.ev(e.a, module, frame, env2, nxt, primed)) or True)
            return frozenset(out)
        if k == "tuple":
            return Fn({i + 1: self.ev(x, module, frame, env, nxt, primed) for i, x in enumerate(e.a)})
        if k == "record":
            return Fn({name: self.ev(x, module, frame, env, nxt, primed) for name, x in e.a})
        if k == "recset":
            return RecordSet([(name, self.ev(x, module, frame, env, nxt, primed)) for name, x in e.a])
        if k == "funcset":
            return FuncSet(self.ev(e.a, module, frame, env, nxt, primed), self.ev(e.b, module, frame, env, nxt, primed))
        if k == "fcons":
            d = {}

            def add(env2):
                keys = [env2[n] for names, _ in e.a for n in names]
                d[keys[0] if len(keys) == 1 else Fn({i + 1: x for i, x in enumerate(keys)})] = \
                    self.ev(e.b, module, frame, env2, nxt, primed)
                return True
            self.each_binding(e.a, 0, 0, module, frame, env, nxt, primed, add)
            return Fn(d)
        if k == "except":
            f = self.ev(e.a, module, frame, env, nxt, primed)
            for path, rhs in e.b:
                f = self.except_update(f, path, 0, rhs, module, frame, env, nxt, primed, e)
            return f
        if k == "at":
            if "@" not in env:
                raise TlaEvalError(f"@ outside EXCEPT (line {e.line})")
            return env["@"]
        if k == "unchanged":
            names = self.unchanged_vars(e.a, module, frame, env)
            return all(values_equal(self.read_next(n, nxt, e), self.cur[n]) for n in names)
        raise TlaEvalError(f"cannot evaluate a {k} expression (line {e.line})")

    def read_next(self, name, nxt, e):
        if nxt is None or name not in nxt:
            raise TlaEvalError(f"{name}' read before it is assigned (line {e.line})")
        return nxt[name]

    def let_env(self, e, module, frame, env):
        env2 = dict(env)
        for d in e.a:
            env2[d.name] = LetOp(d, module, frame, env2) if d.params else Thunk(d.body, module, frame, env2)
            env2 = dict(env2)  # later definitions see earlier ones; each closure keeps the scope it was defined in
        return env2

    def except_update(self, f, path, i, rhs, module, frame, env, nxt, primed, node):
        if i == len(path):
            return self.ev(rhs, module, frame, {**env, "@": f}, nxt, primed)
        if not isinstance(f, Fn):
            raise TlaEvalError(f"EXCEPT on a non-function {f!r} (line {node.line})")
        kind, what = path[i]
        if kind == "fld":
            key = what
        else:
            ks = [self.ev(x, module, frame, env, nxt, primed) for x in what]
            key = ks[0] if len(ks) == 1 else Fn({j + 1: x for j, x in enumerate(ks)})
        if key not in f.d:
            raise TlaEvalError(f"EXCEPT at {key!r}, outside the domain of {f!r} (line {node.line})")  # TLC: a warning
        d = dict(f.d)
        d[key] = self.except_update(f.d[key], path, i + 1, rhs, module, frame, env, nxt, primed, node)
        return Fn(d)

    def each_binding(self, binders, bi, ni, module, frame, env, nxt, primed, fn):
        """Calls fn(env') for every binding of the binders in canonical order until fn returns False."""
        if bi == len(binders):
            return fn(env)
        names, sexpr = binders[bi]
        s = self.ev(sexpr, module, frame, env, nxt, primed)
        nb, nn = (bi, ni + 1) if ni + 1 < len(names) else (bi + 1, 0)
        for x in set_elements(s):
            if not self.each_binding(binders, nb, nn, module, frame, {**env, names[ni]: x}, nxt, primed, fn):
                return False
        return True

    def ev_quant(self, e, bi, ni, module, frame, env, nxt, primed):
        exists = e.a == "E"
        found = [not exists]

        def body(env2):
            b = self.ev_bool(e.c, module, frame, env2, nxt, primed)
            if b == exists:
                found[0] = exists
                return False
            return True
        self.each_binding(e.b, 0, 0, module, frame, env, nxt, primed, body)
        return found[0]

    def ev_op(self, e, module, frame, env, nxt, primed):
        name, args = e.a, e.b
        if name == "implies":
            return (not self.ev_bool(args[0], module, frame, env, nxt, primed)) or \
                self.ev_bool(args[1], module, frame, env, nxt, primed)
        if name == "not":
            return not self.ev_bool(args[0], module, frame, env, nxt, primed)
        a = self.ev(args[0], module, frame, env, nxt, primed)
        if name == "neg":
            return -self.int_(a, e)
        if name == "powerset":
            return PowerSet(a)
        if name == "domain":
            if not isinstance(a, Fn):
                raise TlaEvalError(f"DOMAIN of a non-function (line {e.line})")
            return frozenset(a.d)
        if name == "bigunion":
            out = set()
            for s in set_elements(a):
                out |= as_frozenset(s)
            return frozenset(out)
        b = self.ev(args[1], module, frame, env, nxt, primed)
        if name == "eq":
            return values_equal(a, b)
        if name == "ne":
            return not values_equal(a, b)
        if name == "in":
            return set_contains(b, a)
        if name == "notin":
            return not set_contains(b, a)
        if name in ("lt", "gt", "le", "ge", "plus", "minus", "times", "div", "range"):
            x, y = self.int_(a, e), self.int_(b, e)
            if name == "lt":
                return x < y
            if name == "gt":
                return x > y
            if name == "le":
                return x <= y
            if name == "ge":
                return x >= y
            if name == "plus":
                return x + y
            if name == "minus":
                return x - y
            if name == "times":
                return x * y
            if name == "div":
                return x // y
            return frozenset(range(x, y + 1))
        if name == "union":
            return as_frozenset(a) | as_frozenset(b)
        if name == "cap":
            if isinstance(a, LazySet) and not isinstance(b, LazySet):
                a, b = b, a
            return frozenset(x for x in set_elements(a) if set_contains(b, x))
        if name == "setminus":
            return frozenset(x for x in set_elements(a) if not set_contains(b, x))
        if name == "subseteq":
            return all(set_contains(b, x) for x in set_elements(a))
        if name == "equiv":
            return self.bool_(a, e) == self.bool_(b, e)
        raise TlaEvalError(f"operator {name} is outside the subset (line {e.line})")

    @staticmethod
    def int_(v, e):
        if isinstance(v, int) and not isinstance(v, bool):
            return v
        raise TlaEvalError(f"expected an integer at line {e.line}, got {v!r}")

    @staticmethod
    def bool_(v, e):
        if isinstance(v, bool):
            return v
        raise TlaEvalError(f"expected a boolean at line {e.line}, got {v!r}")

    # ---- actions --------------------------------------------------------------------------
    def unchanged_vars(self, e, module, frame, env):
        """The root variables an UNCHANGED argument names: a variable, a tuple of them, or a definition of one."""
        while e.kind == "paren":
            e = e.a
        if e.kind == "tuple":
            out = []
            for x in e.a:
                out += self.unchanged_vars(x, module, frame, env)
            return out
        v = self.resolve_var(e, module, frame, env)
        if v is not None:
            return [v]
        if e.kind == "ident":
            d = module.visible.get(e.a) if e.a not in env else None
            if isinstance(d, Def) and not d.params:
                return self.unchanged_vars(d.body, d.home, frame, {})
            if frame is not None and e.a not in env and (e.a in module.variables or e.a in module.constants):
                sub = frame.inst.substs.get(e.a) or Node("ident", e.a)
                return self.unchanged_vars(sub, frame.outer_module, frame.outer_frame, {})
        raise TlaEvalError(f"UNCHANGED of something that is not a tuple of variables (line {e.line})")

    def assign_target(self, lhs, module, frame, env):
        """Root variable name when `lhs` is an assignable occurrence: x' (or x while evaluating Init)."""
        while lhs.kind == "paren":
            lhs = lhs.a
        if lhs.kind == "prime":
            return self.resolve_var(lhs.a, module, frame, env)
        if self.init_mode:
            return self.resolve_var(lhs, module, frame, env)
        return None

    def act(self, e: Node, module, frame, env, nxt: dict, label):
        """Generator of the (partial) next-state assignments that satisfy `e`, extending `nxt`."""
        k = e.kind
        if k == "paren":
            yield from self.act(e.a, module, frame, env, nxt, label)
            return
        if k == "and":
            yield from self.act_and(e.a, 0, module, frame, env, nxt, label)
            return
        if k == "or":
            mine = label is not None and label[0] is None
            for i, it in enumerate(e.a):
                if mine:
                    x = it
                    while x.kind == "paren":
                        x = x.a
                    label[0] = x.a if x.kind in ("ident", "apply") else (x.b if x.kind == "inst" else i)
                yield from self.act(it, module, frame, env, nxt, label)
            if mine:
                label[0] = None
            return
        if k == "quant" and e.a == "E":
            yield from self.act_exists(e, 0, 0, module, frame, env, nxt, label)
            return
        if k == "if":
            c = self.ev_bool(e.a, module, frame, env, nxt, False)
            yield from self.act(e.b if c else e.c, module, frame, env, nxt, label)
            return
        if k == "let":
            yield from self.act(e.b, module, frame, self.let_env(e, module, frame, env), nxt, label)
            return
        if k == "ident" and e.a in env and isinstance(env[e.a], Thunk):
            t = env[e.a]
            if self.is_action_shaped(t.expr):
                yield from self.act(t.expr, t.module, t.frame, t.env, nxt, label)
                return
        if k in ("ident", "apply", "inst"):
            op = self.find_operator(e, module, frame, env)
            if op is not None:
                d, home, f2, base, params = op
                args = e.c if k == "inst" else (e.b if k == "apply" else [])
                new_env = self.bind_args(e, params, args or [], base, module, frame, env)
                yield from self.act(d.body, home, f2, new_env, nxt, label)
                return
        if k == "unchanged":
            names = self.unchanged_vars(e.a, module, frame, env)
            new = None
            for n in names:
                if n in nxt or (new is not None and n in new):
                    have = (new if new is not None and n in new else nxt)[n]
                    if not values_equal(have, self.cur[n]):
                        return
                else:
                    if new is None:
                        new = dict(nxt)
                    self.state_reads += 1
                    new[n] = self.cur[n]
            yield nxt if new is None else new
            return
        if k == "op" and e.a in ("eq", "in"):
            target = self.assign_target(e.b[0], module, frame, env)
            if target is not None and target not in nxt:
                rhs = self.ev(e.b[1], module, frame, env, nxt, False)
                if e.a == "eq":
                    yield {**nxt, target: rhs}
                else:
                    for x in set_elements(rhs):
                        yield {**nxt, target: x}
                return
        if self.ev_bool(e, module, frame, env, nxt, False):
            yield nxt

    @staticmethod
    def is_action_shaped(e):
        return e.kind in ("and", "or", "quant", "if", "let", "apply", "inst", "unchanged", "paren", "ident") or \
            (e.kind == "op" and e.a in ("eq", "in"))

    def act_and(self, items, i, module, frame, env, nxt, label):
        if i == len(items):
            yield nxt
            return
        for n1 in self.act(items[i], module, frame, env, nxt, label):
            yield from self.act_and(items, i + 1, module, frame, env, n1, label)

    def act_exists(self, e, bi, ni, module, frame, env, nxt, label):
        binders = e.b
        if bi == len(binders):
            yield from self.act(e.c, module, frame, env, nxt, label)
            return
        names, sexpr = binders[bi]
        s = self.ev(sexpr, module, frame, env, nxt, False)
        nb, nn = (bi, ni + 1) if ni + 1 < len(names) else (bi + 1, 0)
        for x in set_elements(s):
            yield from self.act_exists(e, nb, nn, module, frame, {**env, names[ni]: x}, nxt, label)
