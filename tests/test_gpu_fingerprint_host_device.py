"""kmc_fingerprint<W> (csrc/kmc_common.h) is ONE source for the host and the device: the host computes fingerprints for `contains`,
traces, checkpoints and the owner of Init, the kernels for everything else.  Since round 6 states of eight words and more absorb two
words per 64 x 64 -> 128-bit multiply (`unsigned __int128` under g++ on the host, under hiprtc on the device) and narrower states keep
the per-word chain: both forms, both sides, on real states — the fingerprint the enumerator kernel writes beside every successor
equals the host's for the same words, and the host finds by ITS fingerprint what the search stored by the device's."""
import pytest

from kafka_specification_amd import CheckerConfig, ModelChecker
from kafka_specification_amd.configs import BASELINE_CONFIGS, CONFIG4_DEEP, HEADLINE

pytestmark = pytest.mark.gpu

CASES = [("headline: three words, the per-word chain", HEADLINE, 10),
         ("config 4 at SURVEY's sizing: four words", CONFIG4_DEEP, 5),
         ("config 5: ten words, the folded multiply", BASELINE_CONFIGS["config4_kip320_7brokers_log8"], 5)]


@pytest.mark.parametrize("name,base,levels", CASES, ids=[c[0].split(":")[0] for c in CASES])
def test_host_and_device_compute_the_same_fingerprint_and_find_each_others_states(name, base, levels):
    cfg = CheckerConfig(**{k: v for k, v in base.items() if k != "invariants"}, invariants=("TypeOk",), max_levels=levels,
                        table_capacity=1 << 22, frontier_capacity=1 << 20)
    with ModelChecker(cfg) as mc:
        r = mc.run()
        assert r.verdict == "level_limit" and r.distinct > 200
        last = mc.frontier_states()
        assert len(last) == r.levels[-1]
        step = max(1, len(last) // 60)
        seen_fps, checked = set(), 0
        for row in last[::step]:
            words = [int(x) for x in row]
            assert mc.contains(words), "the host's fingerprint of a stored state does not find it"
            for (t, fp_device, _kind) in mc.successors(words):
                assert mc.fingerprint(t) == fp_device, (name, t)
                seen_fps.add(fp_device)
                checked += 1
        assert checked > 100 and len(seen_fps) > 50
        # a state the search never reached: a successor of the LAST level that is not itself in that level or an earlier one
        stored = {tuple(int(x) for x in row) for row in last}
        absent = 0
        for row in last[::step]:
            for (t, _fp, _k) in mc.successors([int(x) for x in row]):
                if t not in stored and not mc.contains(list(t)):
                    absent += 1
        assert absent > 0, "every successor of the last level was already stored?"
