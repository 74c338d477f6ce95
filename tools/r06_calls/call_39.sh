#!/bin/bash
# round 6, call 39: the default bench line (its timed legs spread their seen-sets: KMC_SEEN_SET_SPREAD=16, set by bench.py) against the
# same line with the chunks as they come (KMC_SEEN_SET_SPREAD=1), as the box's first processes, alternating, twice each.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r06_calls/call_39.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_39; mkdir -p $O
for rep in 1 2; do for sp in 16 1; do
  ( time KMC_SEEN_SET_SPREAD=$sp timeout 900 python bench.py > $O/bench_${sp}_$rep.json 2> $O/bench_${sp}_$rep.err ) 2>&1 | grep real | tr '\n' ' '
  python - $O/bench_${sp}_$rep.json $sp <<'PY' | tee -a $O/summary.txt
import json, sys
j = json.load(open(sys.argv[1])); b = j['config'].get('step_breakdown') or {}
row = ['spread x%s' % sys.argv[2], 'headline %.2f (k_expand %.2f, clear %.2f) open %.2fs golden %s' % (j['ms_per_step'], b.get('k_expand_ms', 0), b.get('clear_seen_set_ms', 0), j['config'].get('open_s') or -1, j['config']['matches_oracle_golden'])]
t = j.get('traces_kept') or {}
row.append('traces %.2f (%.2f)' % (t.get('ms_per_step', 0), t.get('k_expand_ms', 0)))
row.append('orbit %.2f' % (j.get('orbit_counting') or {}).get('ms_per_step', 0))
for k, v in j.get('baseline_configs', {}).items():
    row.append('%s %.2f (%.2f) %s' % (k.split('_')[0] + ('d' if 'deep' in k else ''), v.get('ms_per_step', 0), (v.get('step_breakdown') or {}).get('k_expand_ms', 0), v.get('matches_oracle_golden')))
s = j.get('stretch_1gpu', {}); row.append('stretch %.3f %s' % (s.get('time_to_exhaustive_s', 0), s.get('matches_oracle_golden')))
c = j.get('cold_start') or {}; row.append('cold %.2f / %.2f' % (c.get('wall_s', 0), c.get('wall_s_notrace', 0)))
print(' | '.join(row))
PY
done; done
tail -3 $O/bench_16_1.err
