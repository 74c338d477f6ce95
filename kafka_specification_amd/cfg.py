"""A reader for the subset of TLC's model-configuration (.cfg) syntax these specs need
[TLC-recall]: CONSTANT(S), INIT, NEXT, SPECIFICATION, INVARIANT(S), CHECK_DEADLOCK, and
`\\*` / `(* *)` comments.  SYMMETRY / VIEW / PROPERTY are rejected: each one changes the set of
distinct states (or asks for liveness), which this checker does not do.  CONSTRAINT is accepted in
exactly one form — `CONSTRAINT StateConstraint` with root module MCAsyncIsr (models/MCAsyncIsr.tla),
whose constraint is lowered into the AsyncIsr kernels (AsyncIsr.tla is unbounded without it).

The reference repository ships no .cfg files (its .gitignore excludes *.toolbox), so the
`models/*.cfg` twins in this repo are authored here; names bind to
KafkaReplication.tla:32-36 (constants) and :101,320,334,345 (invariants).
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional

from .checker import CheckerConfig

KEYWORDS = {"CONSTANT", "CONSTANTS", "INIT", "NEXT", "SPECIFICATION", "INVARIANT", "INVARIANTS",
            "CHECK_DEADLOCK", "SYMMETRY", "CONSTRAINT", "CONSTRAINTS", "ACTION_CONSTRAINT", "VIEW",
            "PROPERTY", "PROPERTIES", "ALIAS", "POSTCONDITION"}
UNSUPPORTED = {"SYMMETRY", "ACTION_CONSTRAINT", "VIEW", "PROPERTY", "PROPERTIES", "ALIAS", "POSTCONDITION"}
# root modules whose lowered model carries another name: the MC module adds the state constraint
MODULE_TO_MODEL = {"MCAsyncIsr": "AsyncIsr"}


class CfgError(ValueError):
    pass


@dataclass
class ModelCfg:
    constants: Dict[str, object] = field(default_factory=dict)
    init: Optional[str] = None
    next: Optional[str] = None
    specification: Optional[str] = None
    invariants: List[str] = field(default_factory=list)
    constraints: List[str] = field(default_factory=list)
    check_deadlock: Optional[bool] = None


def _strip_comments(text: str) -> str:
    text = re.sub(r"\(\*.*?\*\)", " ", text, flags=re.S)
    return "\n".join(line.split("\\*")[0] for line in text.splitlines())


def _parse_value(tok: str):
    tok = tok.strip()
    if tok.startswith("{") and tok.endswith("}"):
        inner = tok[1:-1].strip()
        return [] if not inner else [_parse_value(x) for x in inner.split(",")]
    if re.fullmatch(r"-?\d+", tok):
        return int(tok)
    if tok in ("TRUE", "FALSE"):
        return tok == "TRUE"
    if tok.startswith('"') and tok.endswith('"'):
        return tok[1:-1]
    return tok  # a model value


def parse_cfg(text: str) -> ModelCfg:
    cfg = ModelCfg()
    toks = re.findall(r"\{[^}]*\}|\"[^\"]*\"|<-|=|[^\s=]+", _strip_comments(text))
    i, section = 0, None
    while i < len(toks):
        t = toks[i]
        if t in KEYWORDS:
            if t in UNSUPPORTED:
                raise CfgError(f"{t} is not supported (it changes the distinct-state count or asks for liveness)")
            section = t
            i += 1
            continue
        if section in ("CONSTANT", "CONSTANTS"):
            if i + 1 < len(toks) and toks[i + 1] == "<-":
                raise CfgError(f"`{t} <- ...` (substitution by an operator of the spec) is not supported: no TLA+ is parsed, "
                               "so a replaced definition cannot be honoured; assign a value with `=`")
            if i + 2 < len(toks) + 1 and i + 1 < len(toks) and toks[i + 1] == "=":
                if i + 2 >= len(toks):
                    raise CfgError(f"constant {t} has no value")
                cfg.constants[t] = _parse_value(toks[i + 2])
                i += 3
                continue
            raise CfgError(f"expected `{t} = value` in CONSTANTS")
        if section == "INIT":
            cfg.init = t
        elif section == "NEXT":
            cfg.next = t
        elif section == "SPECIFICATION":
            cfg.specification = t
        elif section in ("INVARIANT", "INVARIANTS"):
            cfg.invariants.append(t)
        elif section in ("CONSTRAINT", "CONSTRAINTS"):
            cfg.constraints.append(t)
        elif section == "CHECK_DEADLOCK":
            if t not in ("TRUE", "FALSE"):
                raise CfgError("CHECK_DEADLOCK takes TRUE or FALSE")
            cfg.check_deadlock = t == "TRUE"
        else:
            raise CfgError(f"unexpected token {t!r}")
        i += 1
    return cfg


def to_checker_config(module: str, cfg: ModelCfg, **overrides) -> CheckerConfig:
    """Bind a parsed .cfg to the lowered model of `module` (the root module's name)."""
    from ._native import MODELS, INVARIANTS, ASYNC_INVARIANTS
    if module == "AsyncIsr":
        raise CfgError("AsyncIsr.tla is unbounded (version: Nat, offsets: [Replicas -> Nat]); check it through "
                       "the root module models/MCAsyncIsr.tla, which adds CONSTRAINT StateConstraint")
    model = MODULE_TO_MODEL.get(module, module)
    if model not in MODELS:
        raise CfgError(f"module {module!r} has no lowered model; known: "
                       f"{sorted((set(MODELS) - {'AsyncIsr'}) | set(MODULE_TO_MODEL))}")
    c = cfg.constants
    kw: Dict[str, object] = dict(model=model)
    if cfg.constraints and model != "AsyncIsr":
        raise CfgError("CONSTRAINT is not supported for this module (it changes the distinct-state count)")

    def need(name):
        if name not in c:
            raise CfgError(f"constant {name} is not assigned in the .cfg")
        return c[name]

    def need_int(name):
        v = need(name)
        if isinstance(v, bool) or not isinstance(v, int):
            raise CfgError(f"constant {name} must be an integer, got {v!r}")
        return v

    if module == "IdSequence":
        kw["max_id"] = need_int("MaxId")
        allowed_inv = {"TypeOk"}
    elif model == "AsyncIsr":
        # AsyncIsr.tla:22-29 + MaxVersion / StateConstraint of models/MCAsyncIsr.tla
        reps = need("Replicas")
        if not isinstance(reps, list) or len(set(map(str, reps))) != len(reps):
            raise CfgError("Replicas must be a set of distinct model values")
        if str(need("Leader")) not in map(str, reps):
            raise CfgError("Leader must be an element of Replicas (ASSUME Leader \\in Replicas, AsyncIsr.tla:29)")
        if need_int("MaxOffset") <= 0:
            raise CfgError("MaxOffset must be positive (ASSUME MaxOffset > 0, AsyncIsr.tla:28)")
        if cfg.constraints != ["StateConstraint"]:
            raise CfgError("MCAsyncIsr needs exactly `CONSTRAINT StateConstraint`: AsyncIsr is unbounded without it")
        # the engine numbers the replicas with Leader first; the others keep their order
        kw.update(n_replicas=len(reps), log_size=need_int("MaxOffset"), max_leader_epoch=need_int("MaxVersion"))
        allowed_inv = set(ASYNC_INVARIANTS)
    elif module == "FiniteReplicatedLog":
        reps, recs = need("Replicas"), need("LogRecords")
        if not isinstance(reps, list) or not isinstance(recs, list):
            raise CfgError("Replicas and LogRecords must be sets of model values")
        need("Nil")
        kw.update(n_replicas=len(reps), n_log_records=len(recs), log_size=need_int("LogSize"))
        allowed_inv = {"TypeOk"}
    else:
        reps = need("Replicas")
        if not isinstance(reps, list) or len(set(map(str, reps))) != len(reps):
            raise CfgError("Replicas must be a set of distinct model values")
        if "NONE" in map(str, reps):
            raise CfgError('Replicas must not contain "NONE" (ASSUME None \\notin Replicas, KafkaReplication.tla:42)')
        kw.update(n_replicas=len(reps), log_size=need_int("LogSize"), max_records=need_int("MaxRecords"),
                  max_leader_epoch=need_int("MaxLeaderEpoch"))
        allowed_inv = set(INVARIANTS)
    if cfg.specification is not None and cfg.specification != "Spec":
        raise CfgError("only SPECIFICATION Spec is known")
    if cfg.specification is not None and (cfg.init is not None or cfg.next is not None):
        # [TLC-recall] TLC refuses the combination too (EC.TLC_CONFIG_NOT_BOTH_SPEC_AND_INIT)
        raise CfgError("a .cfg gives either SPECIFICATION or INIT / NEXT, not both")
    if cfg.init not in (None, "Init") or cfg.next not in (None, "Next"):
        raise CfgError("only INIT Init / NEXT Next are known")
    for inv in cfg.invariants:
        if inv not in allowed_inv:
            raise CfgError(f"unknown invariant {inv} for module {module}")
    kw["invariants"] = tuple(cfg.invariants)
    kw["check_deadlock"] = True if cfg.check_deadlock is None else cfg.check_deadlock  # TLC's default is TRUE
    kw.update(overrides)
    return CheckerConfig(**kw)
