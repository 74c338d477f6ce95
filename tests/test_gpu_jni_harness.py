"""GPU: integration/jni/kmcjni.c — the JNI glue a TLC maintainer would add — EXECUTED end to end against libkmc.so, with
tests/jni_stub/fake_jvm.c in the JVM's place (a JNIEnv over a toy object model and a C main that plays
KmcModelChecker.java's callers).  No JVM exists in this image: this runs every line of the C half (Config marshalling, the
per-level Progress call-back, Result objects, TraceState arrays, contains, checkpoint / recover, exceptions); the Java half
is still never compiled.  Checked against the C oracle."""
import json
import os
import subprocess

import pytest

import kmo
from test_jni_shim import build_harness

pytestmark = pytest.mark.gpu
MODEL_ID = {"KafkaTruncateToHighWatermark": 2, "Kip101": 3, "Kip279": 4, "Kip320": 5, "Kip320FirstTry": 6}
INV_INDEX = {"TypeOk": 0, "WeakIsr": 1, "StrongIsr": 2}


def _run(model, N, L, R, E, mask, *extra):
    env = dict(os.environ, KMC_NO_TORCH="1")
    p = subprocess.run([build_harness(), str(MODEL_ID[model]), str(N), str(L), str(R), str(E), str(mask), *extra],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.stdout.strip(), p.stderr[-1500:]
    return p.returncode, json.loads(p.stdout.strip().splitlines()[-1])


def test_an_exhaustive_check_through_the_jni_glue():
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=2, R=2, E=1, invariants=("TypeOk", "WeakIsr", "StrongIsr")))
    rc, r = _run("Kip320", 3, 2, 2, 1, 7)
    assert rc == 0 and "exception" not in r
    assert (r["verdict"], r["violated_invariant"]) == (0, -1)
    assert (r["distinct"], r["generated"], r["depth"], r["queue_left"]) == (o.distinct, o.generated, o.depth, 0)
    assert r["action_generated"][:9] == o.action_generated[:9] and sum(r["action_generated"]) + 1 == r["generated"]
    # Progress.level was called once per BFS level, from inside run(), with the running totals
    assert r["levels_seen_by_progress"] == o.depth and r["last_progress_depth"] == o.depth and r["last_progress_distinct"] == o.distinct
    assert r["contains_init"] == 1 and r["contains_other"] == 0


def test_a_violation_and_its_trace_through_the_jni_glue():
    inv = ("TypeOk", "StrongIsr")
    ocfg = kmo.make_config("Kip101", N=3, L=2, R=2, E=2, invariants=inv)
    o = kmo.Run(ocfg)
    assert o.verdict == "invariant"
    rc, r = _run("Kip101", 3, 2, 2, 2, 1 | 4, "trace")
    assert rc == 0 and (r["verdict"], r["violated_invariant"], r["violation_depth"]) == (1, 2, o.viol_depth)
    assert r["violation_count"][2] == o.viol_count["StrongIsr"]
    tr = [(t["action"], bytes.fromhex(t["canonical"])) for t in r["trace"]]
    assert len(tr) == o.viol_depth and tr[0] == (None, o.state(0))
    from kafka_specification_amd import CheckerConfig, ModelChecker
    with ModelChecker(CheckerConfig(model="Kip101", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=2, device=-1)) as mc:
        names = mc.action_names()
    for (_, prev), (act, cur) in zip(tr, tr[1:]):
        assert (names.index(act), cur) in kmo.successors(ocfg, prev, o.sb)     # TraceState.action is kmc_action_name's string
    assert not kmo.check_invariant(ocfg, INV_INDEX["StrongIsr"], tr[-1][1])


def test_checkpoint_and_recover_through_the_jni_glue(tmp_path):
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=2, R=2, E=1, invariants=("TypeOk",)))
    rc, r = _run("Kip320", 3, 2, 2, 1, 1, "levels=8", f"ckpt={tmp_path / 'jni.ckpt'}")
    assert rc == 0 and (r["verdict"], r["distinct"], r["generated"], r["depth"]) == (0, o.distinct, o.generated, o.depth)
    assert r["resumed_levels"] == o.depth - 8


def test_a_refused_configuration_surfaces_as_a_java_exception():
    rc, r = _run("Kip320", 9, 2, 2, 1, 1)      # nine replicas: kmc_open refuses (N <= 8)
    assert rc == 3 and r["exception"] == "java/lang/IllegalStateException" and r["message"].startswith("kmc_open: ")
