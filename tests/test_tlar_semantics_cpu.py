"""Independent cases for Oracle-R's evaluator (oracle/tlar): tiny modules WRITTEN FOR THIS TEST — none of them from the
reference — whose answers are computed by hand from the definitions of TLA+ (Lamport, *Specifying Systems*, ch. 16-17:
the semantics of the constant operators; ch. 14 / the TLC source for how TLC enumerates an action).  Everything the
fixtures under tests/golden/oracle_r_*.json rest on is the evaluator's reading of the language; here that reading is held
to the book, construct by construct, with no Kafka in sight.

Where the language itself does not decide (how many times TLC *generates* a successor, what it does with a CONSTRAINT) the
case says [TLC-recall]; DESIGN.md section 5 lists every such choice with the test below that fixes it."""
import os
import tempfile

import pytest

from oracle.tlar import Checker
from oracle.tlar.interp import Interp, Loader
from oracle.tlar.values import ModelValue, TlaEvalError


def _interp(body, constants=None, extra_modules=None, name="T"):
    d = tempfile.mkdtemp(prefix="tlar_sem_")
    for n, src in (extra_modules or {}).items():
        open(os.path.join(d, n + ".tla"), "w").write(src)
    open(os.path.join(d, name + ".tla"), "w").write(f"---- MODULE {name} ----\n{body}\n====\n")
    return Interp(Loader([d]), name, constants or {}), d


def holds(body, name="P", constants=None, extra_modules=None, state=None):
    ip, _ = _interp(body, constants, extra_modules)
    return ip.holds(state or {}, name)


def run(body, constants=None, extra_modules=None, name="T", **kw):
    _, d = _interp(body, constants, extra_modules, name)
    return Checker(name, constants or {}, [d]).run(**kw)


# ---------------------------------------------------------------------------------------------------------------
# constant operators: functions, records, tuples (Specifying Systems 16.1.7-16.1.9)
# ---------------------------------------------------------------------------------------------------------------
def test_except_replaces_one_point_and_at_is_the_old_value():
    assert holds("""EXTENDS Integers
f == [x \\in 1..3 |-> x * 10]
P == /\\ [f EXCEPT ![2] = 7] = [x \\in 1..3 |-> IF x = 2 THEN 7 ELSE x * 10]
     /\\ [f EXCEPT ![2] = @ + 1][2] = 21
     /\\ [f EXCEPT ![2] = @ + 1][1] = 10
     /\\ DOMAIN [f EXCEPT ![2] = 0] = 1..3""")


def test_except_on_nested_records_and_functions():
    # [f EXCEPT ![a].fld = e] = [f EXCEPT ![a] = [@ EXCEPT !.fld = e]]  (16.1.7)
    assert holds("""EXTENDS Integers
f == [r \\in {"p", "q"} |-> [hw |-> 0, log |-> [o \\in 0..1 |-> -1]]]
g == [f EXCEPT !["p"].hw = 5, !["q"].log[1] = 9]
P == /\\ g["p"].hw = 5 /\\ g["q"].hw = 0
     /\\ g["q"].log[1] = 9 /\\ g["q"].log[0] = -1 /\\ g["p"].log[1] = -1
     /\\ g = [f EXCEPT !["p"] = [@ EXCEPT !.hw = 5], !["q"] = [@ EXCEPT !.log = [@ EXCEPT ![1] = 9]]]
     /\\ [f EXCEPT !["p"].hw = @ + 1]["p"].hw = 1""")


def test_except_clauses_apply_left_to_right():
    # [f EXCEPT !a = e1, !a = e2]: the second sees the result of the first (16.1.7: defined by iteration)
    assert holds("""EXTENDS Integers
f == [x \\in {1} |-> 0]
P == [f EXCEPT ![1] = 2, ![1] = @ + 1][1] = 3""")


def test_a_record_is_a_function_of_strings_and_a_tuple_a_function_of_1_to_n():
    assert holds("""EXTENDS Integers
P == /\\ [a |-> 1, b |-> 2] = [x \\in {"a", "b"} |-> IF x = "a" THEN 1 ELSE 2]
     /\\ [a |-> 1, b |-> 2].b = 2 /\\ [a |-> 1, b |-> 2]["a"] = 1
     /\\ DOMAIN [a |-> 1, b |-> 2] = {"a", "b"}
     /\\ <<7, 8, 9>>[2] = 8 /\\ DOMAIN <<7, 8, 9>> = 1..3
     /\\ <<7, 8>> = [i \\in {1, 2} |-> i + 6]
     /\\ [a |-> 1] # [a |-> 2] /\\ [a |-> 1] # [b |-> 1] /\\ <<1, 2>> # <<2, 1>>""")


def test_function_applied_outside_its_domain_is_an_error_not_a_value():
    with pytest.raises(TlaEvalError):
        holds("EXTENDS Integers\nf == [x \\in 1..2 |-> x]\nP == f[3] = 3")
    with pytest.raises(TlaEvalError):
        holds('P == [a |-> 1].b = 1')


def test_function_sets_and_their_membership():
    # [S -> T] = the set of all functions with domain S and values in T (16.1.7)
    assert holds("""EXTENDS Integers
F == [{1, 2} -> {0, 1}]
P == /\\ [x \\in {1, 2} |-> 0] \\in F
     /\\ [x \\in {1, 2} |-> x - 1] \\in F
     /\\ [x \\in {1, 2} |-> x] \\notin F
     /\\ [x \\in {1} |-> 0] \\notin F
     /\\ [x \\in {1, 2, 3} |-> 0] \\notin F
     /\\ \\A f \\in F : f[1] \\in {0, 1} /\\ DOMAIN f = {1, 2}
     /\\ {f[1] * 2 + f[2] : f \\in F} = 0..3
     /\\ [{} -> {1}] = {[x \\in {} |-> 1]}
     /\\ [{1} -> {}] = {}""")


def test_record_sets_and_their_membership():
    assert holds("""EXTENDS Integers
R == [id : 0..1, epoch : {5}]
P == /\\ [id |-> 1, epoch |-> 5] \\in R
     /\\ [id |-> 2, epoch |-> 5] \\notin R
     /\\ [id |-> 1] \\notin R
     /\\ [id |-> 1, epoch |-> 5, x |-> 0] \\notin R
     /\\ R = {[id |-> 0, epoch |-> 5], [id |-> 1, epoch |-> 5]}
     /\\ {r.id : r \\in R} = {0, 1}""")


# ---------------------------------------------------------------------------------------------------------------
# sets (16.1.6), SUBSET / UNION, quantifiers over the empty set, CHOOSE (16.1.2)
# ---------------------------------------------------------------------------------------------------------------
def test_powerset_membership_and_enumeration():
    assert holds("""EXTENDS Integers
P == /\\ {1} \\in SUBSET {1, 2} /\\ {} \\in SUBSET {1, 2} /\\ {1, 2} \\in SUBSET {1, 2}
     /\\ {3} \\notin SUBSET {1, 2} /\\ {1, 3} \\notin SUBSET {1, 2}
     /\\ 1 \\notin SUBSET {1, 2}
     /\\ SUBSET {} = {{}}
     /\\ SUBSET {1, 2} = {{}, {1}, {2}, {1, 2}}
     /\\ {s \\in SUBSET {1, 2, 3} : 2 \\in s} = {{2}, {1, 2}, {2, 3}, {1, 2, 3}}""")


def test_set_operators():
    assert holds("""EXTENDS Integers
P == /\\ {1, 2} \\union {2, 3} = {1, 2, 3} /\\ {1, 2} \\cup {} = {2, 1}
     /\\ {1, 2} \\cap {2, 3} = {2} /\\ {1, 2} \\ {2, 3} = {1}
     /\\ {1} \\subseteq {1, 2} /\\ ~({1, 3} \\subseteq {1, 2}) /\\ {} \\subseteq {}
     /\\ UNION {{1}, {2, 3}, {}} = {1, 2, 3} /\\ UNION {} = {}
     /\\ {x \\in 1..5 : x > 3} = {4, 5} /\\ {x * x : x \\in 1..3} = {1, 4, 9}
     /\\ {1, 1, 2} = {2, 1}
     /\\ 1..0 = {} /\\ 2..2 = {2} /\\ 3 \\in 1..3 /\\ 4 \\notin 1..3
     /\\ 3 \\in Nat /\\ 0 \\in Nat /\\ -1 \\notin Nat /\\ -1 \\in Int""")


def test_quantifiers_over_the_empty_set():
    assert holds("P == /\\ ~(\\E x \\in {} : TRUE) /\\ (\\A x \\in {} : FALSE)")


def test_quantifiers_with_several_bound_variables():
    assert holds("""EXTENDS Integers
P == /\\ \\E x, y \\in 1..3 : x + y = 6
     /\\ ~(\\E x, y \\in 1..3 : x + y = 7)
     /\\ \\A x \\in 1..2, y \\in 3..4 : x < y
     /\\ \\E x \\in 1..2, y \\in {x + 1} : y = 3""")


def test_choose_picks_the_only_witness_is_deterministic_and_refuses_an_empty_choice():
    assert holds("""EXTENDS Integers
Max(S) == CHOOSE m \\in S : \\A v \\in S : m >= v
Min(S) == CHOOSE m \\in S : \\A v \\in S : m <= v
P == /\\ (CHOOSE x \\in {7} : TRUE) = 7
     /\\ (CHOOSE x \\in 1..9 : x * x = 49) = 7
     /\\ Max({3, 9, 4}) = 9 /\\ Min({3, 9, 4}) = 3 /\\ Max({5}) = 5
     /\\ (CHOOSE x \\in 1..9 : x > 4) = (CHOOSE x \\in 1..9 : x > 4)
     /\\ (CHOOSE x \\in 1..9 : x > 4) \\in 5..9""")
    with pytest.raises(TlaEvalError):
        holds("EXTENDS Integers\nP == (CHOOSE x \\in 1..3 : x > 5) = 1")
    with pytest.raises(TlaEvalError):
        holds("P == (CHOOSE x \\in {} : TRUE) = 1")


# ---------------------------------------------------------------------------------------------------------------
# logic, arithmetic, precedence
# ---------------------------------------------------------------------------------------------------------------
def test_boolean_operators():
    assert holds("""P == /\\ (FALSE => FALSE) /\\ (FALSE => TRUE) /\\ (TRUE => TRUE) /\\ ~(TRUE => FALSE)
     /\\ (TRUE <=> TRUE) /\\ (FALSE <=> FALSE) /\\ ~(TRUE <=> FALSE) /\\ (TRUE \\equiv TRUE)
     /\\ (TRUE \\/ FALSE) /\\ ~(FALSE \\/ FALSE) /\\ ~(TRUE /\\ FALSE) /\\ \\lnot FALSE
     /\\ (IF TRUE THEN 1 ELSE 2) = 1 /\\ (IF FALSE THEN 1 ELSE 2) = 2""")


def test_a_non_boolean_where_a_boolean_is_needed_is_an_error():
    with pytest.raises(TlaEvalError):
        holds("P == 1 /\\ TRUE")
    with pytest.raises(TlaEvalError):
        holds("P == IF 3 THEN TRUE ELSE FALSE")


def test_arithmetic_and_operator_precedence():
    assert holds("""EXTENDS Integers
P == /\\ 1 + 2 * 3 = 7 /\\ (1 + 2) * 3 = 9 /\\ 7 - 2 - 1 = 4 /\\ 7 \\div 2 = 3 /\\ -3 + 5 = 2
     /\\ 1..2 + 1 = {1, 2, 3}
     /\\ 2 \\in {1} \\union {2}
     /\\ {1} \\union {2} \\subseteq 1..2
     /\\ 3 \\geq 3 /\\ 3 \\leq 3 /\\ 3 >= 2 /\\ 2 <= 3 /\\ 2 < 3 /\\ 3 > 2 /\\ 2 # 3 /\\ 2 /= 3
     /\\ ~ 1 = 2
     /\\ 1 + 1 = 2 /\\ 2 = 1 + 1""")


def test_strings_and_model_values_equal_only_themselves():
    a, b = ModelValue("a"), ModelValue("b")
    assert holds("""CONSTANTS A, B, S
P == /\\ "NONE" = "NONE" /\\ "NONE" # "none"
     /\\ A = A /\\ A # B /\\ A \\in S /\\ {A, B} = S /\\ "NONE" \\notin S
     /\\ [r \\in S |-> 0][A] = 0""", constants=dict(A=a, B=b, S=frozenset((a, b))))


# ---------------------------------------------------------------------------------------------------------------
# definitions: parameters, LET, laziness, EXTENDS / INSTANCE / LOCAL (ch. 17)
# ---------------------------------------------------------------------------------------------------------------
def test_operators_with_parameters_and_let():
    assert holds("""EXTENDS Integers
Add(a, b) == a + b
Twice(x) == Add(x, x)
P == /\\ Twice(4) = 8
     /\\ (LET y == 3 z == y + 1 IN y * z) = 12
     /\\ (LET Sq(v) == v * v IN Sq(3) + Sq(4)) = 25""")


def test_let_definitions_and_arguments_are_evaluated_only_when_used():
    # TLC evaluates LET definitions and operator arguments lazily [TLC-recall]; the reference relies on it where a LET
    # binds an expression that is only defined behind a guard
    assert holds("""EXTENDS Integers
f == [x \\in 1..2 |-> x]
Pick(c, a, b) == IF c THEN a ELSE b
P == /\\ (LET bad == f[99] IN IF TRUE THEN 1 ELSE bad) = 1
     /\\ Pick(TRUE, 1, f[99]) = 1
     /\\ (FALSE /\\ f[99] = 1) = FALSE
     /\\ (TRUE \\/ f[99] = 1)
     /\\ (FALSE => f[99] = 1)""")


COUNTER = """---- MODULE Counter ----
EXTENDS Integers
CONSTANT Limit
VARIABLE c
LOCAL Hidden == 41
Visible == Hidden + 1
Inc == c < Limit /\\ c' = c + 1
IsZero == c = 0
AtMost(n) == c <= n
====
"""


def test_instance_with_substitutes_expressions_for_constants_and_variables():
    r = run("""EXTENDS Integers
VARIABLES x, other
C == INSTANCE Counter WITH c <- x, Limit <- 2 + 1
Init == C!IsZero /\\ other = 0
Next == C!Inc /\\ UNCHANGED other
Inv == C!AtMost(3) /\\ C!Visible = 42""", extra_modules={"Counter": COUNTER}, invariants=("Inv",))
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"]) == (4, 4, 4, "ok")   # x = 0, 1, 2, 3


def test_instance_without_with_substitutes_the_symbols_of_the_same_name():
    r = run("""EXTENDS Integers
CONSTANT Limit
VARIABLE c
C == INSTANCE Counter
Init == c = 1
Next == C!Inc
Inv == c <= Limit""", constants=dict(Limit=2), extra_modules={"Counter": COUNTER}, invariants=("Inv",))
    assert (r["distinct"], r["levels"]) == (2, [1, 1])


def test_two_instances_of_one_module_keep_their_own_substitutions():
    # KafkaReplication.tla:77-78 instantiates IdSequence twice
    r = run("""EXTENDS Integers
VARIABLES a, b
A == INSTANCE Counter WITH c <- a, Limit <- 1
B == INSTANCE Counter WITH c <- b, Limit <- 2
Init == a = 0 /\\ b = 0
Next == (A!Inc /\\ UNCHANGED b) \\/ (B!Inc /\\ UNCHANGED a)""", extra_modules={"Counter": COUNTER})
    assert r["distinct"] == 2 * 3 and r["generated"] == 1 + 7     # edges of the 2 x 3 grid: 3 + 4


def test_local_definitions_are_not_exported_but_work_inside_and_in_a_root_module():
    with pytest.raises(Exception):
        holds("""EXTENDS Integers
VARIABLE x
C == INSTANCE Counter WITH c <- x, Limit <- 1
P == C!Hidden = 41""", extra_modules={"Counter": COUNTER}, state={"x": 0})
    with pytest.raises(Exception):
        holds("EXTENDS Lib\nP == Secret = 1", extra_modules={"Lib": "---- MODULE Lib ----\nLOCAL Secret == 1\nOpen == Secret + 1\n====\n"})
    assert holds("EXTENDS Integers, Lib\nP == Open = 2", extra_modules={"Lib": "---- MODULE Lib ----\nEXTENDS Integers\nLOCAL Secret == 1\nOpen == Secret + 1\n====\n"})
    # a LOCAL Next of the ROOT module is what the .cfg's NEXT names (Kip101.tla:49, Kip279.tla:53)
    r = run("""EXTENDS Integers
VARIABLE x
Init == x = 0
LOCAL Next == x < 2 /\\ x' = x + 1""")
    assert r["distinct"] == 3


def test_extends_imports_definitions_transitively():
    mods = {"A": "---- MODULE A ----\nEXTENDS Integers\nOne == 1\n====\n", "B": "---- MODULE B ----\nEXTENDS A\nTwo == One + 1\n====\n"}
    assert holds("EXTENDS B\nP == Two = 2 /\\ One = 1", extra_modules=mods)


def test_comments_line_block_and_nested():
    assert holds("""EXTENDS Integers
\\* a line comment with == and /\\ in it
(* a block comment
   (* nested *) still a comment *)
P == 1 = 1 \\* trailing""")


# ---------------------------------------------------------------------------------------------------------------
# actions: how TLC enumerates the successors of a state
# ---------------------------------------------------------------------------------------------------------------
def test_one_successor_per_satisfying_binding_of_a_bounded_exists():
    # two bindings, the same successor: generated twice, found once [TLC-recall: "states generated" counts both]
    r = run("""EXTENDS Integers
VARIABLE x
Init == x = 0
Next == \\E i \\in {1, 2} : x < 1 /\\ x' = 1""")
    assert (r["distinct"], r["generated"], r["levels"]) == (2, 1 + 2, [1, 1])


def test_every_disjunct_that_holds_yields_its_own_successor_even_when_state_level():
    # [TLC-recall] a disjunction inside an action is split whether or not its disjuncts mention primed variables:
    # both (x = 0) and (x < 5) hold, the one successor is generated twice (Kip320.tla:82-83 is this shape)
    r = run("""EXTENDS Integers
VARIABLE x
Init == x = 0
Next == /\\ x < 1
        /\\ \\/ x = 0
           \\/ x < 5
        /\\ x' = 1""")
    assert (r["distinct"], r["generated"]) == (2, 1 + 2)
    # ... while inside a VALUE (here: the IF's condition) a disjunction is just a boolean
    r = run("""EXTENDS Integers
VARIABLE x
Init == x = 0
Next == x < 1 /\\ x' = (IF x = 0 \\/ x < 5 THEN 1 ELSE 2)""")
    assert (r["distinct"], r["generated"]) == (2, 1 + 1)


def test_primed_membership_enumerates_and_later_conjuncts_filter():
    r = run("""EXTENDS Integers
VARIABLE x
Init == x \\in {0, 1}
Next == x < 2 /\\ x' \\in 1..4 /\\ x' > x /\\ x' < 4""")
    # Init: two states, both counted as generated.  0 -> {1,2,3}, 1 -> {2,3}; 2 and 3 have no successor (x < 2 fails)
    assert (r["distinct"], r["generated"], r["levels"]) == (4, 2 + 3 + 2, [2, 2])


def test_unchanged_of_a_tuple_a_nested_tuple_and_a_defined_tuple():
    r = run("""EXTENDS Integers
VARIABLES a, b, c
vars == <<a, b, c>>
rest == <<b, c>>
Init == a = 0 /\\ b = 5 /\\ c = 6
Step == a < 2 /\\ a' = a + 1 /\\ UNCHANGED <<b, c>>
Step2 == a = 2 /\\ a' = 3 /\\ UNCHANGED rest
Step3 == a = 3 /\\ a' = 4 /\\ UNCHANGED <<b, <<c>>>>
Stutter == a = 4 /\\ UNCHANGED vars
Next == Step \\/ Step2 \\/ Step3 \\/ Stutter
Inv == b = 5 /\\ c = 6""", invariants=("Inv",))
    assert (r["distinct"], r["generated"], r["verdict"]) == (5, 1 + 5, "ok")


def test_unchanged_after_an_assignment_is_a_condition():
    r = run("""EXTENDS Integers
VARIABLE x
Init == x = 0
Next == x' \\in {0, 1} /\\ UNCHANGED x""")
    assert (r["distinct"], r["generated"]) == (1, 1 + 1)     # only x' = 0 survives


def test_a_variable_left_unassigned_is_an_error():
    with pytest.raises(TlaEvalError):
        run("""EXTENDS Integers
VARIABLES x, y
Init == x = 0 /\\ y = 0
Next == x' = 1""")


def test_except_on_a_function_valued_variable_and_a_set_valued_one():
    r = run("""EXTENDS Integers
VARIABLES f, reqs
Init == f = [r \\in {"p", "q"} |-> 0] /\\ reqs = {}
Bump(r) == f[r] < 1 /\\ f' = [f EXCEPT ![r] = @ + 1] /\\ reqs' = reqs \\union {[who |-> r, n |-> f[r]]}
Next == \\E r \\in {"p", "q"} : Bump(r)
Inv == \\A m \\in reqs : m.n = 0""", invariants=("Inv",))
    # f: (0,0) -> (1,0),(0,1) -> (1,1); reqs follows f
    assert (r["distinct"], r["generated"], r["levels"], r["verdict"]) == (4, 1 + 4, [1, 2, 1], "ok")


def test_if_and_let_inside_an_action():
    r = run("""EXTENDS Integers
VARIABLE x
Init == x = 0
Next == LET y == x + 1 IN IF y < 3 THEN x' = y ELSE x' \\in {0, 7} /\\ x < 7""")
    # 0 -> 1 -> 2 -> {0, 7}; 7 -> IF 8 < 3 .. ELSE (x' \in {0,7} /\ 7 < 7): nothing
    assert (r["distinct"], r["generated"]) == (4, 1 + 1 + 1 + 2)


def test_action_generated_is_keyed_by_the_disjuncts_of_next():
    r = run("""EXTENDS Integers
VARIABLE x
Init == x = 0
Up == x < 2 /\\ x' = x + 1
Reset(v) == x = 2 /\\ x' = v
Next == \\/ Up
        \\/ \\E v \\in {0, 1} : Reset(v)""")
    assert r["action_generated"] == {"Up": 2, 1: 2} or r["action_generated"] == {"Up": 2, "Reset": 2}
    assert r["distinct"] == 3 and r["generated"] == 5


def test_invariant_violation_depth_deadlock_and_constraint():
    body = """EXTENDS Integers
VARIABLE x
Init == x = 0
Next == x < 5 /\\ x' = x + 1
Small == x < 3
Bound == x <= 3"""
    r = run(body, invariants=("Small",))
    assert r["verdict"] == "invariant" and r["violation"]["depth"] == 4 and len(r["violation"]["trace"]) == 4
    assert run(body, check_deadlock=True)["verdict"] == "deadlock"
    assert run(body)["verdict"] == "ok" and run(body)["deadlock_states"] == 1
    # [TLC-recall] CONSTRAINT: a successor outside it is generated (counted) and checked, never stored or explored
    r = run(body, constraint="Bound", invariants=("Small",), stop_on_violation=False)
    assert (r["distinct"], r["generated"]) == (4, 1 + 4) and r["outside_violations"] == {"Small": 1}


def test_assume_is_checked():
    with pytest.raises(TlaEvalError):
        run("""EXTENDS Integers
CONSTANT K
ASSUME K \\in Nat
VARIABLE x
Init == x = 0
Next == x' = x""", constants=dict(K=-1))
