#!/bin/bash
# round 3, symmetry call 5: sparse tiles — the orbit-counting suite, the headline, BASELINE config 4
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/s5
timeout 900 python -m pytest tests/test_gpu_symmetry.py -x -q -n 4 > gpurun_out/s5/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s5/tests.log
tail -5 gpurun_out/s5/tests.log
export KMC_NO_TORCH=1
timeout 200 python tools/sym_headline.py 3 sym 2>&1 | grep ms_step | tail -2
KMC_BENCH_TABLE=$((1<<28)) KMC_BENCH_FRONTIER=$((1<<25)) timeout 600 python bench.py --workload Kip279,5,2,2,1 --no-cpu-baseline > gpurun_out/s5/bench_config4.json 2> gpurun_out/s5/bench_config4.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/s5/bench_config4.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["config"]["distinct_states"], d["roofline"]["kernel_seconds_per_step"])
o=d.get("orbit_counting"); print({k:o[k] for k in ("value","ms_per_step","speedup_over_plain","stored_states","every_count_equals_the_plain_run","kernel_seconds_per_step")})
PY
