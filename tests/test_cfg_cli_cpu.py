"""Host logic that needs no GPU: the .cfg reader, its binding to the lowered models, the state
pretty-printer and the CLI's error paths."""
import glob
import os

import pytest

from kafka_specification_amd.cfg import CfgError, parse_cfg, to_checker_config
from kafka_specification_amd.format import format_state
from kafka_specification_amd import tlc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_headline_cfg_binds_to_the_headline_config():
    from kafka_specification_amd.configs import HEADLINE
    c = to_checker_config("Kip320", parse_cfg(open(os.path.join(ROOT, "models", "Kip320.cfg")).read()))
    assert (c.model, c.n_replicas, c.log_size, c.max_records, c.max_leader_epoch) == (
        HEADLINE["model"], HEADLINE["n_replicas"], HEADLINE["log_size"], HEADLINE["max_records"],
        HEADLINE["max_leader_epoch"])
    assert tuple(c.invariants) == tuple(HEADLINE["invariants"]) and c.check_deadlock is False


def test_every_shipped_cfg_parses_and_binds():
    for path in glob.glob(os.path.join(ROOT, "models", "*.cfg")):
        name = os.path.splitext(os.path.basename(path))[0]
        module = {"Kip279_5brokers": "Kip279", "Kip320_7brokers": "Kip320", "LeaderInIsr": "Kip320",
                  "KafkaTruncateToHighWatermark_3brokers": "KafkaTruncateToHighWatermark",
                  "MCAsyncIsr_small": "MCAsyncIsr", "MCAsyncIsr_outside": "MCAsyncIsr"}.get(name, name)
        c = to_checker_config(module, parse_cfg(open(path).read()))
        c.to_native()


def test_cfg_syntax():
    m = parse_cfg('''
        (* block
           comment *)
        CONSTANT MaxId = 7   \\* trailing comment
        SPECIFICATION Spec
        INVARIANT TypeOk
    ''')
    assert m.constants == {"MaxId": 7} and m.specification == "Spec" and m.invariants == ["TypeOk"]
    c = to_checker_config("IdSequence", m)
    assert c.max_id == 7 and c.check_deadlock is True  # TLC checks deadlock unless told otherwise
    m = parse_cfg("CONSTANTS Replicas = {a, b}\n LogRecords = {x}\n Nil = nil\n LogSize = 3\nCHECK_DEADLOCK FALSE")
    c = to_checker_config("FiniteReplicatedLog", m)
    assert (c.n_replicas, c.n_log_records, c.log_size, c.check_deadlock) == (2, 1, 3, False)


@pytest.mark.parametrize("text,module", [
    ("SYMMETRY Perms\nCONSTANT MaxId = 1", "IdSequence"),                      # changes the distinct-state count
    ("CONSTANTS Replicas = {b1, NONE}\nLogSize=1\nMaxRecords=1\nMaxLeaderEpoch=1", "Kip320"),  # ASSUME :42
    ("CONSTANTS Replicas = {b1, b2}\nLogSize=1\nMaxRecords=1", "Kip320"),     # MaxLeaderEpoch missing
    ("CONSTANT MaxId = 1\nINVARIANT StrongIsr", "IdSequence"),
    ("CONSTANT MaxId = 1", "AsyncIsr"),                                        # unbounded without MCAsyncIsr's constraint
    ("CONSTANT MaxId = 1", "KafkaReplication"),                                # has no Next: no lowered model
    ("CONSTANT MaxId = 1\nPROPERTY Live", "IdSequence"),
    ("CONSTANT MaxId <- SmallId", "IdSequence"),                               # substitution: nothing is parsed to honour it
    ("CONSTANT MaxId = many", "IdSequence"),                                   # a model value where an integer is needed
    ("CONSTANTS Replicas = {b1, b2}\nLogSize=two\nMaxRecords=1\nMaxLeaderEpoch=1", "Kip320"),
    ("CONSTANT MaxId = 1\nSPECIFICATION Spec\nINIT Init\nNEXT Next", "IdSequence"),   # TLC refuses both, too
])
def test_cfg_rejections(text, module):
    with pytest.raises(CfgError):
        to_checker_config(module, parse_cfg(text))


def test_format_state_prints_tla_values():
    from kafka_specification_amd import CheckerConfig
    cfg = CheckerConfig(model="Kip320", n_replicas=2, log_size=2, max_records=2, max_leader_epoch=1)
    # b1: end 1 hw 0 epoch 0 leader b1 isr {b1,b2} records <<[id 0, epoch 0], Nil>>; b2: empty follower of nobody
    b = bytes([1, 0, 1, 1, 3, 1, 0]) + bytes([0, 0, 0, 0, 0, 0, 0]) + bytes([1, 1, 1, 1, 3, 1, 3, 0, 0])
    s = format_state(cfg, b)
    assert "/\\ nextRecordId = 1" in s and "/\\ nextLeaderEpoch = 1" in s
    assert 'b1 :> [hw |-> 0, leaderEpoch |-> 0, leader |-> b1, isr |-> {b1, b2}]' in s
    assert 'b2 :> [hw |-> 0, leaderEpoch |-> -1, leader |-> "NONE", isr |-> {}]' in s
    assert "records |-> <<[id |-> 0, epoch |-> 0], -1>>" in s
    assert "leaderAndIsrRequests = {[leaderEpoch |-> 0, leader |-> b1, isr |-> {b1, b2}]}" in s
    assert "quorumState = [leaderEpoch |-> 0, leader |-> b1, isr |-> {b1, b2}]" in s


def test_cli_error_paths(capsys):
    assert tlc.main([os.path.join(ROOT, "models", "Nope.tla")]) == 2          # no cfg
    import torch
    if not torch.cuda.is_available():
        rc = tlc.main([os.path.join(ROOT, "models", "IdSequence.tla"), "-deadlock"])
        assert rc == 3 and "no CPU fallback" in capsys.readouterr().err         # fails loudly without a GPU


def test_native_cli_error_paths(tmp_path):
    import subprocess
    exe = os.path.join(ROOT, "kafka_specification_amd", "tlc")
    assert os.path.exists(exe), "native CLI not built (python __graft_entry__.py)"
    r = subprocess.run([exe, os.path.join(ROOT, "models", "Nope.tla")], capture_output=True, text=True)
    assert r.returncode == 2 and "not found" in r.stderr
    bad = tmp_path / "x.cfg"
    bad.write_text("CONSTANT MaxId = 3\nSYMMETRY Perms\n")
    r = subprocess.run([exe, "-config", str(bad), os.path.join(ROOT, "models", "IdSequence.tla")],
                       capture_output=True, text=True)
    assert r.returncode == 2 and "SYMMETRY is not supported" in r.stderr
    for text, msg in (("CONSTANT MaxId <- SmallId\n", "is not supported"), ("CONSTANT MaxId = many\n", "must be an integer"),
                      ("CONSTANT MaxId = 3\nSPECIFICATION Spec\nINIT Init\nNEXT Next\n", "not both")):
        bad.write_text(text)
        r = subprocess.run([exe, "-config", str(bad), os.path.join(ROOT, "models", "IdSequence.tla")],
                           capture_output=True, text=True)
        assert r.returncode == 2 and msg in r.stderr, (text, r.stderr)
    r = subprocess.run([exe, os.path.join(ROOT, "models", "AsyncIsr.tla"), "-config",
                        os.path.join(ROOT, "models", "IdSequence.cfg")], capture_output=True, text=True)
    assert r.returncode == 2 and "unbounded" in r.stderr               # needs MCAsyncIsr's CONSTRAINT
    r = subprocess.run([exe, os.path.join(ROOT, "models", "KafkaReplication.tla"), "-config",
                        os.path.join(ROOT, "models", "IdSequence.cfg")], capture_output=True, text=True)
    assert r.returncode == 2 and "no lowered model" in r.stderr        # KafkaReplication.tla has no Next
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, os.path.join(ROOT, "models", "IdSequence.tla"), "-deadlock"],
                           capture_output=True, text=True)
        assert r.returncode == 3 and "no CPU fallback" in r.stderr  # fails loudly without a GPU


def _both_clis(args):
    """(returncode, stderr) of the Python and the native front end."""
    import subprocess
    import sys
    exe = os.path.join(ROOT, "kafka_specification_amd", "tlc")
    env = dict(os.environ, PYTHONPATH=ROOT)
    py = subprocess.run([sys.executable, "-m", "kafka_specification_amd.tlc"] + args, capture_output=True, text=True, env=env)
    cc = subprocess.run([exe] + args, capture_output=True, text=True)
    return (py.returncode, py.stderr), (cc.returncode, cc.stderr)


def test_spec_identity_guard_refuses_an_edited_spec(tmp_path):
    """The front ends map the module NAME to a lowered model and never parse TLA+, so they hash the spec they are
    given: a Kip320.tla that is not the revision the kernels were lowered from is refused (nothing
    is checked), -force checks the built-in lowering anyway, and a missing file only warns (this repository ships
    no copy of the reference's modules)."""
    import hashlib
    import shutil
    import torch
    no_gpu = not torch.cuda.is_available()
    spec = tmp_path / "Kip320.tla"
    spec.write_text("---- MODULE Kip320 ----\nEXTENDS Integers\nNext == FALSE\n====\n")
    shutil.copy(os.path.join(ROOT, "models", "Kip320.cfg"), tmp_path / "Kip320.cfg")
    digest = hashlib.sha256(spec.read_bytes()).hexdigest()[:16]
    for rc, err in _both_clis([str(spec), "-deadlock", "-table", "1024", "-frontier", "1024"]):
        assert rc == 2 and "differs from the revision" in err and "nothing was checked" in err
        assert digest in err                       # both front ends hash the same bytes (the C++ one with its own SHA-256)
    if no_gpu:
        for rc, err in _both_clis([str(spec), "-deadlock", "-force", "-table", "1024", "-frontier", "1024"]):
            assert rc == 3 and "-force" in err and "no CPU fallback" in err      # past the guard, then no device
        # a module file that does not exist: a warning, then the built-in lowering
        for rc, err in _both_clis([str(tmp_path / "missing" / "Kip320.tla"), "-config", str(tmp_path / "Kip320.cfg"), "-deadlock"]):
            assert rc == 3 and "does not exist" in err


def test_spec_identity_guard_accepts_the_reference_revision(tmp_path):
    """With the real modules beside the .cfg the guard walks EXTENDS / INSTANCE (Kip320 -> Kip279 -> KafkaReplication ->
    Util, IdSequence, FiniteReplicatedLog) and accepts them; one changed byte anywhere in that closure is refused."""
    import shutil
    ref = os.environ.get("KMC_REFERENCE", "/root/reference")
    if not os.path.exists(os.path.join(ref, "KafkaReplication.tla")):
        pytest.skip("the reference's .tla files are not on this box")
    from kafka_specification_amd.spec_revision import check_spec
    for f in os.listdir(ref):
        if f.endswith(".tla"):
            shutil.copy(os.path.join(ref, f), tmp_path / f)
    shutil.copy(os.path.join(ROOT, "models", "MCAsyncIsr.tla"), tmp_path / "MCAsyncIsr.tla")
    for mod in ("Kip320", "Kip320FirstTry", "Kip101", "KafkaTruncateToHighWatermark", "IdSequence", "MCAsyncIsr"):
        assert check_spec(str(tmp_path / f"{mod}.tla")) == ("ok", [])
    shutil.copy(os.path.join(ROOT, "models", "Kip320.cfg"), tmp_path / "Kip320.cfg")
    with open(tmp_path / "FiniteReplicatedLog.tla", "a") as f:
        f.write("\n")                                  # deep in Kip320's closure, through an INSTANCE
    assert check_spec(str(tmp_path / "Kip320.tla"))[0] == "mismatch"
    for rc, err in _both_clis([str(tmp_path / "Kip320.tla"), "-deadlock"]):
        assert rc == 2 and "module FiniteReplicatedLog" in err


def test_collision_estimate_and_tlc_log_parser():
    """The summary prints TLC's estimate of a silent fingerprint collision; tools/tlc_log_diff.py reads the same
    message formats back (it is what tools/verify_with_tlc.sh uses on a real TLC log)."""
    import sys
    lines = tlc.collision_report(279753922, 901914892)
    assert "calculated (optimistic):  val = 9.44E-03" in lines[1] and "2.12E-03" in lines[2]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import tlc_log_diff
    ok = ("Model checking completed. No error has been found.\n"
          "901914892 states generated, 279753922 distinct states found, 0 states left on queue.\n"
          "The depth of the complete state graph search is 46.\n")
    assert tlc_log_diff.parse(ok) == dict(verdict="ok", invariant=None, trace_length=0, generated=901914892,
                                          distinct=279753922, left=0, depth=46)
    bad = ("Error: Invariant StrongIsr is violated.\nError: The behavior up to this point is:\n"
           "State 1: <Initial predicate>\n/\\ x = 1\n\nState 2: <Next line 3, col 1 to line 4, col 2 of module M>\n/\\ x = 2\n\n"
           "663643 states generated, 171601 distinct states found, 90210 states left on queue.\n")
    p = tlc_log_diff.parse(bad)
    assert (p["verdict"], p["invariant"], p["trace_length"], p["left"]) == ("invariant", "StrongIsr", 2, 90210)


def test_stock_tlc_switches_are_ignored_with_a_note_or_refused_never_misread():
    """A wrapper script written for `java tlc2.TLC` passes switches this engine has no use for.  Those that do not change
    what is checked are accepted and ignored (the run goes on: without a GPU it then ends at the device, status 3);
    those that ask for another mode of operation end the run with status 2 and the reason.  Both front ends alike."""
    spec, cfg = os.path.join(ROOT, "models", "IdSequence.tla"), os.path.join(ROOT, "models", "IdSequence.cfg")
    import torch
    has_gpu = torch.cuda.is_available()
    harmless = ["-modelcheck", "-cleanup", "-nowarning", "-coverage", "1", "-checkpoint", "0", "-workers", "auto", "-metadir", "/tmp/x"]
    for (rc, err) in _both_clis(harmless + ["-config", cfg, "-force", "-deadlock", spec]):
        assert rc == (0 if has_gpu else 3), err
        assert err.count("accepted") >= 5 and "unknown option" not in err and "unrecognized" not in err
    for flag, why in (("-simulate", "another mode"), ("-dump", "stay on the GPU"), ("-view", "distinct-state count")):
        for (rc, err) in _both_clis([flag, "-config", cfg, "-force", spec]):
            assert rc == 2 and f"{flag} is not supported" in err and why in err, (flag, err)
