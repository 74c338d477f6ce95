#!/bin/bash
# round 3, final call 2: the representative among the SORTED images at five and six replicas (KmcSymm::canon_sorted) —
# the orbit-counting suite, BASELINE config 4 with and without orbit counting, then (the device sources changed) the profile
# passes of call_final1 again, the bench line quoting them, and the whole suite
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/f2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_symmetry.py -q -n 4 > $O/tests_sym.log 2>&1; echo "sym tests rc=$?" >> $O/tests_sym.log
tail -4 $O/tests_sym.log
KMC_BENCH_TABLE=$((1<<28)) KMC_BENCH_FRONTIER=$((1<<25)) timeout 600 python bench.py --workload Kip279,5,2,2,1 --no-cpu-baseline --steps 3 > $O/bench_config4.json 2> $O/bench_config4.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/f2/bench_config4.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["config"]["distinct_states"], d["roofline"]["kernel_seconds_per_step"])
o=d.get("orbit_counting"); print({k:o[k] for k in ("value","ms_per_step","speedup_over_plain","stored_states","every_count_equals_the_plain_run","kernel_seconds_per_step")})
PY
bash tools/profile.sh r03g > $O/profile_plain.log 2>&1; tail -2 $O/profile_plain.log
PROFILE_BENCH_ARGS=--symmetry bash tools/profile.sh r03g_sym > $O/profile_sym.log 2>&1; tail -2 $O/profile_sym.log
cp gpurun_out/prof_r03g/pmc_summary.json profiles/r03_pmc_summary.json; cp gpurun_out/prof_r03g/summary.json profiles/r03_summary.json
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout 840 python -m pytest tests -q -m gpu -n 4 > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -E " passed| failed|rc=|FAILED|ERROR" $O/tests.log | tail -12
