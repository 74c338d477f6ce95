#!/bin/bash
# round 6, call 42: the last tree as the driver runs it (after the CPU suite learnt to spread over workers: the -m gpu suite must stay serial) - the -m gpu suite, smoke, the default bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_42; mkdir -p $O
ls kafka_specification_amd/kmc_cache | sort > $O/cache_before.txt
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
ls kafka_specification_amd/kmc_cache | sort > $O/cache_after.txt
echo "specialised on the box:"; comm -13 $O/cache_before.txt $O/cache_after.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
j = json.load(open('gpurun_out/r06_42/bench.json'))
print('headline', round(j['ms_per_step'], 2), j['config']['matches_oracle_golden'], round(j['roofline']['frac'], 4), j['roofline'].get('traffic') is not None)
for k, v in j.get('baseline_configs', {}).items():
    print(k, round(v.get('ms_per_step', 0), 2), v.get('matches_oracle_golden'), (v.get('roofline') or {}).get('traffic') is not None, v.get('error'))
s = j.get('stretch_1gpu', {})
print('stretch', s.get('time_to_exhaustive_s'), s.get('matches_oracle_golden'), (s.get('roofline') or {}).get('traffic') is not None, s.get('error'))
PY
python - <<'PY'
import json
j = json.load(open('gpurun_out/r06_42/bench.json'))
print('breakdown', j['config'].get('step_breakdown'))
print('traces_kept', j.get('traces_kept'))
print('cold_start', json.dumps(j.get('cold_start'))[:600])
PY
