#!/bin/bash
# round 6, call 28: the evidence on the LAST tree (after the chunked seen-set and the paired slots; call 10 was the same on the tree before them) - the whole -m gpu suite as the driver runs it (with what the box had to specialise
# itself listed: build() should leave nothing), smoke, rocprofv3 summaries (kernel trace + the PMC passes + a marker trace with the
# roctx ranges) for every kernel a line quotes - the headline, orbit counting, BASELINE configs 4 and 5 (with its kmc_inv row),
# config 4 at SURVEY's sizing, the 6.45 G-state stretch - and, last, so that it finds the summaries of its own machine code, the
# default bench line.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/r06_calls/call_28.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_28; mkdir -p $O
ls kafka_specification_amd/kmc_cache | sort > $O/cache_before.txt
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
ls kafka_specification_amd/kmc_cache | sort > $O/cache_after.txt
echo "specialised on the box:"; comm -13 $O/cache_before.txt $O/cache_after.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
export KMC_NO_TORCH=1
bash tools/profile.sh r06 > $O/profile_plain.log 2>&1; tail -1 $O/profile_plain.log; grep "roctx ranges" $O/profile_plain.log
PROFILE_BENCH_ARGS=--symmetry bash tools/profile.sh r06_sym > $O/profile_sym.log 2>&1; tail -1 $O/profile_sym.log
PROFILE_BENCH_ARGS="--workload Kip279,5,2,2,1" bash tools/profile.sh r06_config4 > $O/profile_config4.log 2>&1; tail -1 $O/profile_config4.log
( export KMC_BENCH_TABLE=$((7<<28)) KMC_BENCH_FRONTIER=$((1<<29))
  PROFILE_BENCH_ARGS="--workload Kip320,7,8,8,3 --level-budget 10" bash tools/profile.sh r06_config5 > $O/profile_config5.log 2>&1; tail -1 $O/profile_config5.log )
( export KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<28))
  PROFILE_BENCH_ARGS="--workload Kip279,5,4,4,3 --level-budget 12" bash tools/profile.sh r06_config4_deep > $O/profile_config4_deep.log 2>&1; tail -1 $O/profile_config4_deep.log )
( export KMC_BENCH_FP128=1 KMC_BENCH_TABLE=15000000000 KMC_BENCH_FRONTIER=600000000 PROFILE_PMC_TIMEOUT=240
  PROFILE_BENCH_ARGS="--workload Kip320,3,6,6,3" bash tools/profile.sh r06_stretch > $O/profile_stretch.log 2>&1; tail -1 $O/profile_stretch.log )
unset KMC_NO_TORCH
# the bench reads profiles/rNN_*pmc_summary.json: put this call's summaries where it looks, then run it as the driver does
for t in r06 r06_sym r06_config4 r06_config5 r06_config4_deep r06_stretch; do for f in kernel_stats.csv summary.json pmc_summary.json; do cp gpurun_out/prof_$t/$f profiles/${t}_$f 2>/dev/null; done; done
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -3; cut -c1-300 $O/bench.json
python - <<'PY' 2>&1 | tee $O/bench_digest.txt
import json
j = json.load(open('gpurun_out/r06_28/bench.json'))
print('headline ms', j['ms_per_step'], j['config'].get('step_breakdown'), j['config']['matches_oracle_golden'], 'frac', j['roofline']['frac'], 'traffic', j['roofline'].get('traffic'))
oc = j.get('orbit_counting') or {}
print('orbit counting', oc.get('ms_per_step'), oc.get('matches_oracle_golden'))
for k, v in j.get('baseline_configs', {}).items():
    print(k, v.get('ms_per_step'), v.get('step_breakdown'), v.get('matches_oracle_golden'), 'frac', (v.get('roofline') or {}).get('frac'), 'traffic', (v.get('roofline') or {}).get('traffic'), 'k_inv', (v.get('k_inv') or {}).get('achieved'), (v.get('k_inv') or {}).get('traffic'), 'cpu', (v.get('cpu_baseline') or {}).get('value'), v.get('error'))
s = j.get('stretch_1gpu', {})
print('stretch', {k: s.get(k) for k in ('time_to_exhaustive_s', 'first_run_wall_s', 'open_s', 'matches_oracle_golden', 'table_slots', 'table_load_at_end', 'error')}, (s.get('roofline') or {}).get('frac'), (s.get('roofline') or {}).get('probes_per_s'), (s.get('roofline') or {}).get('traffic'))
print('cpu', j.get('cpu_baseline', {}).get('value'), 'cold', (j.get('cold_start') or {}).get('wall_s'))
PY
tail -5 $O/bench.err
