"""`tlc2.TLC`-shaped command line [TLC-recall]:

    python -m kafka_specification_amd.tlc [-config X.cfg] [-deadlock] [-continue] [-workers N]
                                          [-fp SEED] [-fp128] [-symmetry] [-fpcheck] [-verify] [-force] [-levels-csv FILE] [-gpus P] [-table SLOTS]
                                          [-frontier STATES] Spec.tla

Maps the root module's name to its lowered GPU model, reads constants / invariants from the
.cfg (default: Spec.cfg next to the module), runs the exhaustive search on the GPU and prints
TLC-style progress and summary lines.  -workers is accepted for command-line compatibility
and ignored (the GPU's waves are the workers).  No TLA+ is parsed: only the modules of
hachikuji/kafka-specification that have a Next are known — so the spec file given on the command line (and
every module it EXTENDS / INSTANCEs, when found beside it) is hashed against the revision the kernels were
lowered from (spec_revision.py); an edited spec is refused unless -force.

The seen-set holds 64-bit fingerprints, like TLC's FPSet: two distinct states with one fingerprint lose a state
silently.  Like TLC, the summary prints the estimated probability of that; -fpcheck runs the search a second
time with another fingerprint seed and compares the counts (a collision moves with the seed).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

from .cfg import CfgError, parse_cfg, to_checker_config
from .checker import ModelChecker
from .format import format_state


def _now():
    return time.strftime("%Y-%m-%d %H:%M:%S")


def collision_report(distinct: int, generated: int, wide: bool = False):
    """TLC's closing estimate [TLC-recall: "calculated (optimistic)" = distinct x (generated - distinct) / 2^64], plus
    the birthday bound n^2 / 2^65 for the fingerprints that were stored.  With -fp128 every entry also holds a second,
    independent 64-bit hash of the state: two states are only merged when both words agree."""
    if wide:
        return ["The seen-set stores 128 bits per state (the fingerprint and an independent check word); probability that two "
                "distinct states were merged:",
                f"  birthday bound on the stored entries:  val = {distinct * distinct / 2.0 ** 129:.2E}"]
    opt = distinct * max(generated - distinct, 0) / 2.0 ** 64
    birthday = distinct * distinct / 2.0 ** 65
    out = ["The seen-set stores 64-bit fingerprints; estimates of the probability that not all reachable states were "
           "checked because two distinct states had the same fingerprint:",
           f"  calculated (optimistic):  val = {opt:.2E}",
           f"  birthday bound on the stored fingerprints:  val = {birthday:.2E}"]
    if birthday > FP128_ADVICE_ABOVE:
        # (Kip320 3/6/6/3: 6,452,700,520 states, birthday bound 1.1 — the 64-bit search returns one state fewer)
        out.append(f"  Recommendation: that bound is above {FP128_ADVICE_ABOVE}: counts of this size are only bit-exact with 128-bit "
                   "entries - re-run with -fp128 (or, on the Kafka modules, -symmetry: a sixth of the stored fingerprints at three brokers).")
    return out


def uncertified(distinct: int, wide: bool) -> bool:
    """A finished 64-bit search whose birthday bound exceeds FP128_ADVICE_ABOVE: its distinct-state count is more likely to be
    off than not (BASELINE's target is the bit-identical count) — both front ends then exit with UNCERTIFIED_EXIT_CODE."""
    return (not wide) and distinct * distinct / 2.0 ** 65 > FP128_ADVICE_ABOVE


UNCERTIFIED_EXIT_CODE = 14


FP128_ADVICE_ABOVE = 0.1


TLC_IGNORED_FLAGS = {
    "-modelcheck": "model checking is the only mode", "-cleanup": "no states directory is written",
    "-nowarning": "no TLA+ is evaluated, so no evaluation warnings exist", "-terse": "values are printed in full",
    "-tool": "no tool-mode message codes", "-gzip": "checkpoints are not compressed", "-debug": "no debug output",
    "-noGenerateSpecTE": "no trace-expression spec is generated", "-difftrace": "traces print every variable of every state",
}
TLC_IGNORED_WITH_VALUE = {
    "-metadir": "nothing is written there", "-userFile": "the lowered models print nothing",
    "-fpmem": "the fingerprint table lives in HBM: -table SLOTS sizes it", "-fpbits": "one table, no partitioning by bits",
    "-maxSetSize": "no set is enumerated at run time", "-coverage": "action coverage is not collected",
    "-lncheck": "no liveness checking",
}
TLC_REFUSED_FLAGS = {
    "-simulate": "random simulation is another mode of TLC; only exhaustive breadth-first model checking is implemented",
    "-depth": "it belongs to -simulate", "-seed": "it belongs to -simulate", "-aril": "it belongs to -simulate",
    "-dump": "the reachable states stay on the GPU (kmc_frontier_states gives a level's states through the C ABI)",
    "-view": "a VIEW changes the distinct-state count", "-dfid": "depth-first iterative deepening is another search order",
    "-generateSpecTE": "no trace-expression spec is generated",
}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="tlc", add_help=True, prefix_chars="-")
    ap.add_argument("spec")
    ap.add_argument("-config", default=None)
    ap.add_argument("-deadlock", action="store_true", help="do NOT check for deadlock (TLC semantics of -deadlock)")
    ap.add_argument("-continue", dest="cont", action="store_true")
    ap.add_argument("-workers", default="1")
    ap.add_argument("-fp", type=int, default=0)
    ap.add_argument("-gpus", type=int, default=1, help="P > 1: P logical shards on this process's GPU (loopback).  Real "
                    "multi-GPU runs: `python -m torch.distributed.run --nproc-per-node P -m kafka_specification_amd.tlc ...` "
                    "— one rank per GPU, the fingerprint space sharded over the ranks, rank 0 prints")
    ap.add_argument("-table", type=int, default=0)
    ap.add_argument("-frontier", type=int, default=0)
    ap.add_argument("-device", type=int, default=0)
    ap.add_argument("-notrace", action="store_true", help="do not keep predecessor links (no counterexample trace)")
    ap.add_argument("-levels-csv", dest="levels_csv", default=None, metavar="FILE",
                    help="write one line per BFS level (one GPU): frontier expanded, new states, probes, deadlocks, table load, "
                         "k_expand milliseconds, successors generated per disjunct of Next")
    ap.add_argument("-fp128", action="store_true",
                    help="128-bit seen-set entries: the fingerprint plus an independent 64-bit check word per state, in the same "
                         "cache line (no extra memory traffic per probe, twice the table bytes); a 64-bit fingerprint collision "
                         "is then recognised instead of losing a state")
    ap.add_argument("-symmetry", action="store_true",
                    help="orbit counting: the specs never tell two members of Replicas apart, so one state per orbit of the "
                         "permutations of Replicas is stored and expanded and every count is weighted by its orbit's size — the "
                         "numbers printed are those of the plain search (TLC without a SYMMETRY set; TLC's own SYMMETRY prints "
                         "the reduced counts) from ~1/|Replicas|! of the work.  Kafka family / FiniteReplicatedLog, up to 7 "
                         "replicas, one GPU")
    ap.add_argument("-fpcheck", action="store_true",
                    help="run the search again with a second fingerprint seed and compare the counts")
    ap.add_argument("-maxlevels", type=int, default=0, help="stop after this many BFS levels (verdict level_limit)")
    ap.add_argument("-checkpoint", default=None, metavar="DIR",
                    help="with -gpus P / torch.distributed.run and -maxlevels: every shard saves its table and frontier "
                         "there when the level limit is reached (TLC -checkpoint for a sharded search)")
    ap.add_argument("-recover", default=None, metavar="DIR", help="continue a sharded search from such a directory")
    ap.add_argument("-verify", action="store_true",
                    help="differential self-check: a second, differently compiled build of the kernels regenerates every "
                         "level and the per-action / deadlock / violation counts must agree (for constants no oracle reaches)")
    ap.add_argument("-force", action="store_true", help="check the built-in lowering although the spec text differs from it")
    ap.add_argument("-v", action="store_true", dest="verbose",
                    help="one more closing line: where the wall time outside the search went (HIP start-up, code object, allocation, "
                         "first touch of the seen-set: kmc_timing)")
    # Stock TLC's other command-line switches [TLC-recall]: the ones that do not change what is checked are accepted and
    # ignored with a note (a wrapper script written for `java tlc2.TLC` keeps working); the ones that ask for another mode
    # of operation are refused — silently dropping them would answer a different question than the one asked.
    argv = list(sys.argv[1:] if argv is None else argv)
    kept, i = [], 0
    while i < len(argv):
        t = argv[i]
        if t in TLC_IGNORED_FLAGS:
            print(f"Note: {t} is accepted for compatibility and ignored ({TLC_IGNORED_FLAGS[t]})", file=sys.stderr)
        elif t in TLC_IGNORED_WITH_VALUE:
            print(f"Note: {t} {argv[i + 1] if i + 1 < len(argv) else ''} is accepted for compatibility and ignored "
                  f"({TLC_IGNORED_WITH_VALUE[t]})", file=sys.stderr)
            i += 1
        elif t in TLC_REFUSED_FLAGS:
            print(f"Error: {t} is not supported: {TLC_REFUSED_FLAGS[t]}", file=sys.stderr)
            return 2
        elif (t == "-checkpoint" and i + 1 < len(argv) and argv[i + 1].lstrip("-").isdigit()
              and not os.path.isdir(argv[i + 1])):
            # TLC's -checkpoint takes an interval in minutes; here a search takes milliseconds to seconds and -checkpoint DIR
            # names where a level-limited sharded search leaves its state (a directory that exists under a numeric name is
            # a directory; to CREATE one with a numeric name say ./2024)
            print(f"Note: -checkpoint {argv[i + 1]} (TLC's interval in minutes) is accepted and ignored; "
                  "-checkpoint DIR saves a level-limited sharded search", file=sys.stderr)
            i += 1
        else:
            kept.append(t)
        i += 1
    a = ap.parse_args(kept)

    module = os.path.splitext(os.path.basename(a.spec))[0]
    cfg_path = a.config or os.path.splitext(a.spec)[0] + ".cfg"
    if not os.path.exists(cfg_path):
        print(f"Error: configuration file {cfg_path} not found", file=sys.stderr)
        return 2
    try:
        mcfg = parse_cfg(open(cfg_path).read())
        over = dict(hash_seed=a.fp, device=a.device, continue_on_violation=a.cont, keep_trace=not a.notrace,
                    table_capacity=a.table, frontier_capacity=a.frontier, max_levels=a.maxlevels,
                    wide_fingerprint=a.fp128, symmetry=a.symmetry)
        if a.deadlock:
            over["check_deadlock"] = False
        cc = to_checker_config(module, mcfg, **over)
    except CfgError as e:
        print(f"Error: {e}", file=sys.stderr)
        return 2

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        # launched by torch.distributed.run: one shard per rank (RCCL over xGMI, the exchange under the C ABI);
        # every rank computes the same global result, rank 0 reports it
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not dist.is_initialized():
            dist.init_process_group(backend=os.environ.get("KMC_BACKEND", "nccl"))
        if dist.get_backend() == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
        if rank != 0:
            sys.stdout = open(os.devnull, "w")

    if a.verify:
        os.environ["KMC_VERIFY"] = "1"
    from .spec_revision import check_spec
    status, msgs = check_spec(a.spec)
    for m in msgs:
        print(("Warning: " if status != "mismatch" else "Error: ") + m, file=sys.stderr)
    if status == "mismatch":
        if not a.force:
            print("Error: the spec differs from the revision the GPU kernels were lowered from; nothing was checked "
                  "(-force checks the built-in lowering anyway)", file=sys.stderr)
            return 2
        print("Warning: -force: checking the BUILT-IN lowering, not the text of the spec given", file=sys.stderr)

    print(f"kafka_specification_amd model checker (MI355X) — module {module}, config {os.path.basename(cfg_path)}")
    print(f"Running breadth-first search Model-Checking with fp seed {a.fp} on GPU {a.device}.")
    print("Computing initial states...")

    def progress(i):
        if i["depth"] == 1:
            print(f"Finished computing initial states: {i['distinct']} distinct state generated at {_now()}.")
        else:
            print(f"Progress({i['depth']}) at {_now()}: {i['generated']} states generated, "
                  f"{i['distinct']} distinct states found, {i['new_states']} states left on queue.")

    from ._native import KmcError
    from dataclasses import replace

    def search(conf, progress):
        if world > 1:
            from .sharded import check_distributed
            res = check_distributed(conf, progress, a.checkpoint, a.recover)
            return res, res.trace
        if a.gpus > 1:
            from .sharded import check_loopback
            res = check_loopback(conf, a.gpus, a.device, progress, a.checkpoint, a.recover)
            return res, res.trace   # walked owner by owner through the shards' predecessor tables
        with ModelChecker(conf) as mc:
            res = mc.run(progress)
            timing.update(mc.timing())
            nonlocal level_rows
            level_rows = mc.level_stats()
            trace = []
            if res.verdict in ("invariant",) and conf.keep_trace:
                trace = mc.trace()
            elif res.verdict == "deadlock":
                trace = [(None, mc.unpack(mc.witness()))]
            return res, trace

    second = None
    timing = {}
    level_rows = None   # -levels-csv: the per-expansion records of the search on one GPU (kmc_level_stats)
    try:
        res, trace = search(cc, progress)
        if a.fpcheck:
            seed2 = (a.fp * 0x9E3779B97F4A7C15 + 0x5851F42D4C957F2D) & 0xFFFFFFFFFFFFFFFF
            second, _ = search(replace(cc, hash_seed=seed2, keep_trace=False), None)
            second = (seed2, second)
    except KmcError as e:
        print(f"Error: {e}", file=sys.stderr)
        return 3

    rc = 0
    if res.verdict == "ok":
        print("Model checking completed. No error has been found.")
    elif res.verdict == "invariant":
        where = " by the initial state" if res.violation_depth == 1 else ""
        print(f"Error: Invariant {res.violated_invariant} is violated{where}.")
        rc = 12
    elif res.verdict == "deadlock":
        print("Error: Deadlock reached.")
        rc = 11
    else:
        print(f"Error: search stopped: {res.verdict} (table {res.table_capacity} slots, "
              f"frontier {res.frontier_capacity} states)")
        rc = 1
    if trace:
        print("Error: The behavior up to this point is:")
        for k, (act, st) in enumerate(trace, 1):
            head = "<Initial predicate>" if act is None and k == 1 else f"<{act} of module {module}>"
            print(f"State {k}: {head}")
            print(format_state(cc, st))
            print()
    print(f"{res.generated} states generated, {res.distinct} distinct states found, "
          f"{res.queue_left} states left on queue.")
    print(f"The depth of the complete state graph search is {res.depth}.")
    if a.symmetry:
        print(f"Symmetry reduction (orbit counting over the permutations of Replicas): {res.orbit_representatives} states "
              f"were stored and expanded, one per orbit; every count above is the plain search's.")
    # (the seen-set holds the representatives: they are what can collide, and what was probed for)
    stored, probed = res.distinct, res.generated
    if a.symmetry and res.distinct:
        stored = res.orbit_representatives
        probed = res.generated * stored // res.distinct
    for line in collision_report(stored, probed, a.fp128):
        print(line)
    if rc == 0 and uncertified(stored, a.fp128):
        print(f"Warning: the distinct-state count of this run is NOT certified (64-bit fingerprints, birthday bound "
              f"{stored * stored / 2.0 ** 65:.2E} > {FP128_ADVICE_ABOVE}): exit code {UNCERTIFIED_EXIT_CODE}.")
        rc = UNCERTIFIED_EXIT_CODE
    if a.levels_csv and level_rows is not None:
        names = list(res.action_generated)
        with open(a.levels_csv, "w") as lf:
            lf.write("depth,frontier,new_states,stored_new,probes,deadlocks,table_load,expand_ms," + ",".join(names) + "\n")
            for st in level_rows:
                lf.write(f"{st['depth']},{st['frontier']},{st['new_states']},{st['stored_new']},{st['probes']},{st['deadlocks']},"
                         f"{st['table_load']:.6f},{st['expand_ms']:.4f}," + ",".join(str(st['generated'][n]) for n in names) + "\n")
    if second is not None:
        seed2, r2 = second
        same = (r2.verdict, r2.distinct, r2.generated, r2.depth) == (res.verdict, res.distinct, res.generated, res.depth)
        print(f"Fingerprint check: second run with fp seed {seed2}: {r2.generated} states generated, {r2.distinct} distinct "
              f"states found, depth {r2.depth} - " + ("identical to the first run." if same else
              "DIFFERENT from the first run: a fingerprint collision dropped states in at least one of them "
              "(a collision can only lose states: the larger count is the better lower bound)."))
        if not same and rc == 0:
            rc = 13
    print(f"Finished in {res.seconds_total:.3f}s ({res.distinct / max(res.seconds_total, 1e-9):,.0f} distinct states/s; "
          f"{res.seconds_expand:.3f}s in the expand kernel) at ({_now()})")
    if a.verbose and timing:
        print(f"Wall time outside the search: kmc_open {timing['open_s']:.3f}s (HIP initialisation {timing['hip_init_s']:.3f}s, code "
              f"object {timing['code_object_s']:.3f}s, allocation of {timing['device_bytes'] / 2 ** 30:.1f} GiB {timing['alloc_s']:.3f}s), "
              f"first clear of the seen-set {timing['first_clear_s']:.3f}s")
    return rc


if __name__ == "__main__":
    sys.exit(main())
