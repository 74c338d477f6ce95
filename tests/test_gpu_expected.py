"""GPU: every models/*.cfg twin through the .cfg reader and the C ABI against models/EXPECTED.json — the file a JVM
owner diffs stock TLC against (tools/verify_with_tlc.sh).  Both legs of each entry: the default run (stops at the first
violation: verdict, invariant, length of the counterexample) and the exhaustive run (-continue when violated)."""
import json
import os

import pytest

from kafka_specification_amd import ModelChecker
from kafka_specification_amd.cfg import parse_cfg, to_checker_config

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXPECTED = json.load(open(os.path.join(ROOT, "models", "EXPECTED.json")))
CFGS = sorted(k for k, v in EXPECTED.items() if not k.startswith("_") and v.get("exhaustible"))


def _sizes(distinct):
    t = 1 << 22
    while t < 3 * distinct:
        t <<= 1
    return dict(table_capacity=t, frontier_capacity=max(1 << 20, t // 8))


@pytest.mark.parametrize("cfg_name", CFGS)
def test_cfg_twin_matches_expected(cfg_name):
    e = EXPECTED[cfg_name]
    mcfg = parse_cfg(open(os.path.join(ROOT, "models", cfg_name)).read())
    sz = _sizes(e["exhaustive"]["distinct"])
    stop, full = e["stop"], e["exhaustive"]
    with ModelChecker(to_checker_config(e["module"], mcfg, keep_trace=stop["verdict"] == "invariant", **sz)) as mc:
        r = mc.run()
        trace = mc.trace() if r.verdict == "invariant" else []
    assert r.verdict == stop["verdict"] and r.violated_invariant == stop["invariant"]
    if stop["verdict"] == "invariant":
        assert len(trace) == r.violation_depth == stop["trace_length"]
        assert sorted(n for n, c in r.violation_count.items() if c and n in e["invariants"]) == stop["invariants_violated_at_that_depth"]
    if stop["verdict"] == "ok":
        assert (r.distinct, r.generated, r.depth) == (full["distinct"], full["generated"], full["depth"])
    else:
        with ModelChecker(to_checker_config(e["module"], mcfg, continue_on_violation=True, **sz)) as mc:
            c = mc.run()
        assert (c.distinct, c.generated, c.depth, c.queue_left) == (full["distinct"], full["generated"], full["depth"], 0)
        assert (c.verdict, c.violated_invariant, c.violation_depth) == ("invariant", stop["invariant"], stop["trace_length"])
