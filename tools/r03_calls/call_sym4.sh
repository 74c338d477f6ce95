#!/bin/bash
# round 3, symmetry call 4: five and six replicas (the adjacent-transposition walk), BASELINE config 4 with / without symmetry
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/s4
timeout 900 python -m pytest tests/test_gpu_symmetry.py -x -q -k "five_and_six or config4 or finite_replicated or refused" > gpurun_out/s4/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s4/tests.log
tail -15 gpurun_out/s4/tests.log
KMC_BENCH_TABLE=$((1<<28)) KMC_BENCH_FRONTIER=$((1<<25)) timeout 600 python bench.py --workload Kip279,5,2,2,1 --no-cpu-baseline > gpurun_out/s4/bench_config4.json 2> gpurun_out/s4/bench_config4.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/s4/bench_config4.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["config"]["distinct_states"], d["roofline"]["kernel_seconds_per_step"])
print(d.get("orbit_counting"))
PY
tail -3 gpurun_out/s4/bench_config4.err
