"""Oracle-R, part 1: lexer and parser for the TLA+ subset the ten reference modules use.

TEST INFRASTRUCTURE ONLY (see oracle/tlar/__init__.py).  This reads the reference's *text*
(/root/reference/*.tla) — it restates nothing: no operator of the specs is written down here.

Subset (anything else raises TlaSyntaxError — "fail loudly", this is not SANY):
  module header / EXTENDS / CONSTANT(S) / VARIABLE(S) / ASSUME / THEOREM / LOCAL /
  Name == INSTANCE M WITH a <- e, ... / operator definitions with parameters;
  indentation-sensitive /\ and \/ junction lists, \E \A CHOOSE with several binders, LET/IN,
  IF/THEN/ELSE, records, record sets, functions, function sets, EXCEPT with @ and nested paths,
  tuples, set enumeration / filter / map, SUBSET, DOMAIN, UNCHANGED, primes, Inst!Op(args),
  and the temporal forms of the Spec definitions ([]A, [A]_v, SF_v(A), WF_v(A)), which are parsed and never
  evaluated.

Junction lists follow SANY's rule: a list is the run of identical bullets that start in the same column, and
every token of an item lies strictly to the right of that column.
"""
from __future__ import annotations

import re


class TlaSyntaxError(Exception):
    pass


KEYWORDS = {
    "MODULE", "EXTENDS", "CONSTANT", "CONSTANTS", "VARIABLE", "VARIABLES", "ASSUME", "ASSUMPTION", "AXIOM",
    "THEOREM", "INSTANCE", "WITH", "LOCAL", "LET", "IN", "IF", "THEN", "ELSE", "CHOOSE", "SUBSET", "UNION",
    "DOMAIN", "UNCHANGED", "ENABLED", "EXCEPT", "TRUE", "FALSE", "CASE", "OTHER",
}
BACKSLASH_WORDS = {
    "in", "notin", "union", "cup", "cap", "intersect", "subseteq", "leq", "geq", "E", "A", "div", "lnot", "land",
    "lor", "X", "times", "o", "neg", "equiv",
}
# longest first
SYMBOLS = [
    "|->", "<=>", "<-", "->", "==", "=>", "/\\", "\\/", "/=", "<=", "=<", ">=", "<<", ">>", "..", "[]", "<>", "]_",
    ":>", "@@", "=", "#", "<", ">", "+", "-", "*", "'", "!", "@", "[", "]", "(", ")", "{", "}", ",", ":", ".", "~",
]


class Tok:
    __slots__ = ("kind", "text", "line", "col")

    def __init__(self, kind, text, line, col):
        self.kind, self.text, self.line, self.col = kind, text, line, col

    def __repr__(self):
        return f"{self.kind}:{self.text}@{self.line}:{self.col}"


_ID = re.compile(r"[A-Za-z_][A-Za-z0-9_]*")
_NUM = re.compile(r"[0-9]+")


def lex(text: str, fname: str = "?"):
    """Tokens of the first module in `text` (from its ---- MODULE line to its ==== line)."""
    toks = []
    i, n = 0, len(text)
    line, bol = 1, 0  # current line number, index of its first character
    depth = 0         # (* ... *) nesting
    in_module = False
    while i < n:
        ch = text[i]
        if ch == "\n":
            line += 1
            i += 1
            bol = i
            continue
        if depth:
            if text.startswith("(*", i):
                depth += 1
                i += 2
            elif text.startswith("*)", i):
                depth -= 1
                i += 2
            else:
                i += 1
            continue
        if text.startswith("(*", i):
            depth = 1
            i += 2
            continue
        if text.startswith("\\*", i):
            while i < n and text[i] != "\n":
                i += 1
            continue
        if ch in " \t\r":
            i += 1
            continue
        col = i - bol
        if text.startswith("----", i):
            j = i
            while j < n and text[j] == "-":
                j += 1
            toks.append(Tok("SEP", "----", line, col))
            i = j
            continue
        if text.startswith("====", i):
            if in_module:
                toks.append(Tok("END", "====", line, col))
                break
            j = i
            while j < n and text[j] == "=":
                j += 1
            i = j
            continue
        if not in_module:
            # text before the module header is free-form (the licence comment is a (* *) block, but be lenient)
            m = _ID.match(text, i)
            if m and m.group() == "MODULE" and toks and toks[-1].kind == "SEP":
                in_module = True
                toks = [toks[-1], Tok("KW", "MODULE", line, col)]
                i = m.end()
            elif m:
                i = m.end()
            else:
                i += 1
            continue
        if ch == '"':
            j = text.index('"', i + 1)
            toks.append(Tok("STR", text[i + 1:j], line, col))
            i = j + 1
            continue
        m = _NUM.match(text, i)
        if m:
            toks.append(Tok("NUM", m.group(), line, col))
            i = m.end()
            continue
        m = _ID.match(text, i)
        if m:
            w = m.group()
            toks.append(Tok("KW" if w in KEYWORDS else "ID", w, line, col))
            i = m.end()
            continue
        if ch == "\\":
            m = _ID.match(text, i + 1)
            if m and m.group() in BACKSLASH_WORDS:
                toks.append(Tok("OP", "\\" + m.group(), line, col))
                i = m.end()
                continue
            if text.startswith("\\/", i):
                toks.append(Tok("OP", "\\/", line, col))
                i += 2
                continue
            if m:
                raise TlaSyntaxError(f"{fname}:{line}:{col}: unsupported operator \\{m.group()}")
            toks.append(Tok("OP", "\\", line, col))  # set difference
            i += 1
            continue
        for s in SYMBOLS:
            if text.startswith(s, i):
                toks.append(Tok("OP", s, line, col))
                i += len(s)
                break
        else:
            raise TlaSyntaxError(f"{fname}:{line}:{col}: unexpected character {ch!r}")
    if not in_module:
        raise TlaSyntaxError(f"{fname}: no module header found")
    toks.append(Tok("EOF", "", line + 1, -1))
    return toks


# ------------------------------------------------------------------------------------------------
# AST: plain tuples would do, but a tiny node class keeps positions for error messages.
# ------------------------------------------------------------------------------------------------
class Node:
    """kind + fields.  Kinds and their fields:
      num(v) str(v) bool(v) ident(name) at
      op(name, args)            prefix / infix built-in operator, args evaluated by the interpreter
      and(items) or(items)      junction list or infix /\\ \\/ (flattened)
      prime(e)
      apply(name, args)         user / built-in operator applied to arguments: Name(a, b)
      inst(path, name, args)    Inst!Name(args); path = list of instance names
      fapp(f, args)             f[a] / f[a, b]
      dot(e, field)
      tuple(items) setenum(items) setfilter(var, set, pred) setmap(expr, binders)
      quant(q, binders, body)   q in {'E','A'}; binders = [(names, set_expr)]
      choose(var, set, pred)
      let(defs, body)           defs = [Def]
      if(c, a, b)
      fcons(binders, body)      [x \\in S |-> e]
      record(fields)            [(name, expr)]
      recset(fields)            [(name, expr)]
      funcset(dom, rng)
      except(f, updates)        updates = [(path, expr)], path = [('idx', [exprs]) | ('fld', name)]
      unchanged(e) enabled(e)
      temporal(what, ...)       parsed, never evaluated
    """
    __slots__ = ("kind", "a", "b", "c", "line", "col")

    def __init__(self, kind, a=None, b=None, c=None, line=0, col=0):
        self.kind, self.a, self.b, self.c, self.line, self.col = kind, a, b, c, line, col

    def __repr__(self):
        parts = [repr(x) for x in (self.a, self.b, self.c) if x is not None]
        return f"{self.kind}({', '.join(parts)})"


class Def:
    """name(params) == body, defined in module `home` (filled in by the loader)."""
    __slots__ = ("name", "params", "body", "local", "home", "line")

    def __init__(self, name, params, body, local, line):
        self.name, self.params, self.body, self.local, self.home, self.line = name, params, body, local, None, line

    def __repr__(self):
        return f"Def({self.name}/{len(self.params)})"


class InstanceDef:
    """Name == INSTANCE target WITH a <- e, ...   (substitution expressions live in the defining module)."""
    __slots__ = ("name", "target", "substs", "local", "home", "line")

    def __init__(self, name, target, substs, local, line):
        self.name, self.target, self.substs, self.local, self.home, self.line = name, target, substs, local, None, line


class ModuleAst:
    def __init__(self, name):
        self.name = name
        self.extends = []
        self.constants = []
        self.variables = []
        self.assumes = []
        self.theorems = []
        self.defs = []       # Def | InstanceDef, in source order


# binary operators: text -> (left binding power, right binding power, op name)
BINOPS = {
    "=>": (1, 1, "implies"), "<=>": (2, 3, "equiv"), "\\equiv": (2, 3, "equiv"),
    "\\/": (3, 4, "or"), "\\lor": (3, 4, "or"), "/\\": (5, 6, "and"), "\\land": (5, 6, "and"),
    "=": (9, 10, "eq"), "#": (9, 10, "ne"), "/=": (9, 10, "ne"),
    "<": (9, 10, "lt"), ">": (9, 10, "gt"), "<=": (9, 10, "le"), "=<": (9, 10, "le"), "\\leq": (9, 10, "le"),
    ">=": (9, 10, "ge"), "\\geq": (9, 10, "ge"),
    "\\in": (9, 10, "in"), "\\notin": (9, 10, "notin"), "\\subseteq": (9, 10, "subseteq"),
    "\\union": (15, 16, "union"), "\\cup": (15, 16, "union"), "\\cap": (15, 16, "cap"), "\\intersect": (15, 16, "cap"),
    "\\": (15, 16, "setminus"),
    "..": (17, 18, "range"),
    "+": (19, 20, "plus"), "-": (19, 20, "minus"),
    "*": (25, 26, "times"), "\\div": (25, 26, "div"),
}
PREFIX_BP = {"~": 7, "\\lnot": 7, "\\neg": 7, "-": 23, "SUBSET": 15, "UNION": 15, "DOMAIN": 17, "UNCHANGED": 7,
             "ENABLED": 7, "[]": 7, "<>": 7}


class Parser:
    def __init__(self, toks, fname="?"):
        self.toks = toks
        self.i = 0
        self.fname = fname
        self.jstack = []  # columns of the enclosing junction lists

    # ---- token access ---------------------------------------------------------------------
    def raw(self, k=0):
        return self.toks[min(self.i + k, len(self.toks) - 1)]

    def peek(self):
        """Next token, or a STOP token when it lies at or left of the innermost junction column."""
        t = self.toks[self.i]
        if self.jstack and t.kind != "EOF" and t.col <= self.jstack[-1]:
            return Tok("STOP", "", t.line, t.col)
        return t

    def next(self):
        t = self.peek()
        if t.kind == "STOP":
            self.fail("expression ends prematurely (token left of its junction list)", t)
        self.i += 1
        return t

    def fail(self, msg, t=None):
        t = t or self.raw()
        raise TlaSyntaxError(f"{self.fname}:{t.line}:{t.col}: {msg} (at {t.text!r})")

    def at(self, text, kind=None):
        t = self.peek()
        return t.text == text and t.kind in (("OP", "KW") if kind is None else (kind,))

    def accept(self, text):
        if self.at(text):
            self.i += 1
            return True
        return False

    def expect(self, text):
        if not self.accept(text):
            self.fail(f"expected {text!r}")

    def ident(self):
        t = self.next()
        if t.kind != "ID":
            self.fail("expected an identifier", t)
        return t.text

    # ---- module ---------------------------------------------------------------------------
    def module(self):
        t = self.next()
        if t.kind != "SEP":
            self.fail("expected the module header", t)
        self.expect("MODULE")
        m = ModuleAst(self.ident())
        if self.next().kind != "SEP":
            self.fail("expected ---- after the module name")
        while True:
            t = self.peek()
            if t.kind in ("END", "EOF"):
                break
            if t.kind == "SEP":
                self.i += 1
                continue
            if t.kind == "KW" and t.text == "EXTENDS":
                self.i += 1
                m.extends.append(self.ident())
                while self.accept(","):
                    m.extends.append(self.ident())
                continue
            if t.kind == "KW" and t.text in ("CONSTANT", "CONSTANTS", "VARIABLE", "VARIABLES"):
                self.i += 1
                dest = m.constants if t.text.startswith("CONST") else m.variables
                dest.append(self.ident())
                if self.at("("):
                    self.fail("operator-valued constants are outside the subset")
                while self.accept(","):
                    dest.append(self.ident())
                continue
            if t.kind == "KW" and t.text in ("ASSUME", "ASSUMPTION", "AXIOM"):
                self.i += 1
                m.assumes.append(self.expr())
                continue
            if t.kind == "KW" and t.text == "THEOREM":
                self.i += 1
                m.theorems.append(self.expr())
                continue
            local = False
            if t.kind == "KW" and t.text == "LOCAL":
                self.i += 1
                local = True
                t = self.peek()
            if t.kind == "KW" and t.text == "INSTANCE":
                self.fail("unnamed INSTANCE is outside the subset")
            if t.kind != "ID":
                self.fail("expected a definition")
            m.defs.append(self.definition(local))
        return m

    def definition(self, local):
        t = self.next()
        name, params = t.text, []
        if self.accept("("):
            params.append(self.ident())
            while self.accept(","):
                params.append(self.ident())
            self.expect(")")
        self.expect("==")
        if self.at("INSTANCE"):
            self.i += 1
            target = self.ident()
            substs = {}
            if self.accept("WITH"):
                while True:
                    k = self.ident()
                    self.expect("<-")
                    substs[k] = self.expr()
                    if not self.accept(","):
                        break
            if params:
                self.fail("parameterised INSTANCE is outside the subset")
            return InstanceDef(name, target, substs, local, t.line)
        return Def(name, params, self.expr(), local, t.line)

    # ---- expressions ----------------------------------------------------------------------
    def expr(self, min_bp=0):
        left = self.prefix()
        while True:
            t = self.peek()
            if t.kind == "OP" and t.text == "'":
                self.i += 1
                left = Node("prime", left, line=t.line, col=t.col)
                continue
            if t.kind == "OP" and t.text == "[" and self._adjacent_apply_ok(left):
                self.i += 1
                args = [self.expr()]
                while self.accept(","):
                    args.append(self.expr())
                self.expect("]")
                left = Node("fapp", left, args, line=t.line, col=t.col)
                continue
            if t.kind == "OP" and t.text == ".":
                self.i += 1
                left = Node("dot", left, self.ident(), line=t.line, col=t.col)
                continue
            if t.kind != "OP" or t.text not in BINOPS:
                break
            lbp, rbp, name = BINOPS[t.text]
            if lbp < min_bp:
                break
            self.i += 1
            right = self.expr(rbp)
            if name in ("and", "or"):
                items = (left.a if left.kind == name and left.b == "infix" else [left]) + [right]
                left = Node(name, items, "infix", line=t.line, col=t.col)
            else:
                left = Node("op", name, [left, right], line=t.line, col=t.col)
        return left

    @staticmethod
    def _adjacent_apply_ok(left):
        # (a bracketed constructor applied on the spot — [f EXCEPT ![a] = b][c], <<x, y>>[1], [a |-> 1]["a"] — is ordinary
        # function application too; the reference never writes it, tests/test_tlar_semantics_cpu.py does)
        return left.kind in ("ident", "fapp", "dot", "at", "apply", "inst", "prime", "paren", "except", "record", "fcons",
                             "tuple")

    def junction(self, t):
        kind = "and" if t.text in ("/\\", "\\land") else "or"
        col = t.col
        items = []
        while True:
            r = self.raw()
            if not (r.kind == "OP" and r.text == t.text and r.col == col):
                break
            if self.jstack and col <= self.jstack[-1]:
                break
            self.i += 1
            self.jstack.append(col)
            items.append(self.expr())
            self.jstack.pop()
        return Node(kind, items, "list", line=t.line, col=t.col)

    def binders(self):
        """x, y \\in S, z \\in T  ->  [([x, y], S), ([z], T)]"""
        out = []
        while True:
            names = [self.ident()]
            while self.accept(","):
                names.append(self.ident())
            if not self.accept("\\in"):
                self.fail("unbounded quantification is outside the subset")
            out.append((names, self.expr()))
            if not self.accept(","):
                return out

    def prefix(self):
        t = self.peek()
        if t.kind == "STOP" or t.kind in ("EOF", "END", "SEP"):
            self.fail("expected an expression", t)
        if t.kind == "OP" and t.text in ("/\\", "\\/"):
            return self.junction(t)
        self.i += 1
        L = dict(line=t.line, col=t.col)
        if t.kind == "NUM":
            return Node("num", int(t.text), **L)
        if t.kind == "STR":
            return Node("str", t.text, **L)
        if t.kind == "KW":
            if t.text in ("TRUE", "FALSE"):
                return Node("bool", t.text == "TRUE", **L)
            if t.text == "IF":
                c = self.expr()
                self.expect("THEN")
                a = self.expr()
                self.expect("ELSE")
                return Node("if", c, a, self.expr(), **L)
            if t.text == "LET":
                defs = []
                while not self.at("IN"):
                    d = self.definition(False)
                    if isinstance(d, InstanceDef):
                        self.fail("INSTANCE inside LET is outside the subset")
                    defs.append(d)
                self.expect("IN")
                return Node("let", defs, self.expr(), **L)
            if t.text == "CHOOSE":
                v = self.ident()
                if not self.accept("\\in"):
                    self.fail("unbounded CHOOSE is outside the subset")
                s = self.expr()
                self.expect(":")
                return Node("choose", v, s, self.expr(), **L)
            if t.text in PREFIX_BP:
                e = self.expr(PREFIX_BP[t.text])
                name = {"SUBSET": "powerset", "UNION": "bigunion", "DOMAIN": "domain"}.get(t.text)
                if name:
                    return Node("op", name, [e], **L)
                return Node(t.text.lower(), e, **L)
            self.fail("unsupported keyword in an expression", t)
        if t.kind == "ID":
            if t.text.startswith(("SF_", "WF_")) and self.at("("):
                self.i += 1
                a = self.expr()
                self.expect(")")
                return Node("temporal", t.text[:2], t.text[3:], a, **L)
            path = []
            name = t.text
            args = self.call_args()
            while self.at("!"):
                if args:
                    self.fail("parameterised instance prefix is outside the subset")
                self.i += 1
                path.append(name)
                name = self.ident()
                args = self.call_args()
            if path:
                return Node("inst", path, name, args, **L)
            if args:
                return Node("apply", name, args, **L)
            return Node("ident", name, **L)
        # symbols
        if t.text == "@":
            return Node("at", **L)
        if t.text == "(":
            e = self.expr()
            self.expect(")")
            return Node("paren", e, **L)
        if t.text in ("\\E", "\\A"):
            b = self.binders()
            self.expect(":")
            return Node("quant", t.text[1], b, self.expr(), **L)
        if t.text in PREFIX_BP:
            e = self.expr(PREFIX_BP[t.text])
            if t.text in ("~", "\\lnot", "\\neg"):
                return Node("op", "not", [e], **L)
            if t.text == "-":
                return Node("op", "neg", [e], **L)
            return Node("temporal", t.text, e, **L)
        if t.text == "<<":
            items = []
            if not self.at(">>"):
                items.append(self.expr())
                while self.accept(","):
                    items.append(self.expr())
            self.expect(">>")
            return Node("tuple", items, **L)
        if t.text == "{":
            return self.brace(L)
        if t.text == "[":
            return self.bracket(L)
        self.fail("unexpected token in an expression", t)

    def call_args(self):
        # Name(args): the parenthesis must follow on the same junction level; an operator applied to arguments
        if self.at("("):
            self.i += 1
            args = [self.expr()]
            while self.accept(","):
                args.append(self.expr())
            self.expect(")")
            return args
        return []

    def brace(self, L):
        if self.accept("}"):
            return Node("setenum", [], **L)
        first = self.expr()
        if self.accept(":"):
            if first.kind == "op" and first.a == "in" and first.b[0].kind == "ident":
                pred = self.expr()
                self.expect("}")
                return Node("setfilter", first.b[0].a, first.b[1], pred, **L)
            b = self.binders()
            self.expect("}")
            return Node("setmap", first, b, **L)
        items = [first]
        while self.accept(","):
            items.append(self.expr())
        self.expect("}")
        return Node("setenum", items, **L)

    def bracket(self, L):
        r0, r1 = self.raw(0), self.raw(1)
        if r0.kind == "ID" and r1.kind == "OP" and r1.text in ("|->", ":"):
            kind = "record" if r1.text == "|->" else "recset"
            fields = []
            while True:
                k = self.ident()
                self.expect(r1.text)
                fields.append((k, self.expr()))
                if not self.accept(","):
                    break
            self.expect("]")
            return Node(kind, fields, **L)
        if r0.kind == "ID" and r1.kind == "OP" and r1.text in ("\\in", ","):
            # [x \in S |-> e]  (a function constructor); [x \in S -> T] does not occur
            save = self.i
            try:
                b = self.binders()
                self.expect("|->")
            except TlaSyntaxError:
                self.i = save
            else:
                body = self.expr()
                self.expect("]")
                return Node("fcons", b, body, **L)
        e = self.expr()
        if self.accept("EXCEPT"):
            ups = []
            while True:
                self.expect("!")
                path = []
                while not self.at("="):
                    if self.accept("."):
                        path.append(("fld", self.ident()))
                    elif self.accept("["):
                        idx = [self.expr()]
                        while self.accept(","):
                            idx.append(self.expr())
                        self.expect("]")
                        path.append(("idx", idx))
                    else:
                        self.fail("malformed EXCEPT path")
                self.expect("=")
                ups.append((path, self.expr()))
                if not self.accept(","):
                    break
            self.expect("]")
            return Node("except", e, ups, **L)
        if self.accept("->"):
            rng = self.expr()
            self.expect("]")
            return Node("funcset", e, rng, **L)
        if self.accept("]_"):
            sub = self.expr(27)
            return Node("temporal", "box_action", e, sub, **L)
        self.fail("unsupported [ ... ] form")


def parse_module(text: str, fname: str = "?") -> ModuleAst:
    p = Parser(lex(text, fname), fname)
    m = p.module()
    return m
