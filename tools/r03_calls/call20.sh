#!/bin/bash
# round 3, GPU call 20: run-time guards (KmcKafka::guard<K> looped per kind, fused into the kind-major walk) against the
# straight-line block of every instance's guard, on BASELINE configs 4 and 5 and on the headline
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c20; mkdir -p $O; rm -f gpurun_out/sweep.log
export KMC_NO_TORCH=1
tools/sweep.sh "headline_static_guards||" "headline_runtime_guards|-DKMC_RT_GUARDS_MIN_INSTANCES=0|" "headline_static_guards_again||"
cp gpurun_out/sweep.log $O/sweep.log
one() {  # tag, defines, workload args...
  tag=$1; defs=$2; shift 2
  KMC_JIT_DEFINES="$defs" timeout 900 python bench.py "$@" --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    j = json.load(open("$O/$tag.json")); print("$tag", "ms/step %.2f" % j["ms_per_step"], "G/s %.3f" % (j["value"]/1e9), j["config"]["distinct_states"], j["config"]["states_generated"], "k_expand %.2f ms" % (1e3*j["roofline"]["kernel_seconds_per_step"]))
except Exception as e: print("$tag FAILED", e, open("$O/$tag.err").read()[-300:])
PY
}
export KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
one config5_runtime "" --workload Kip320,7,8,8,3 --level-budget 10 --steps 1 --warmup 1
one config5_static "-DKMC_RT_GUARDS_MIN_INSTANCES=100000" --workload Kip320,7,8,8,3 --level-budget 10 --steps 1 --warmup 1
export KMC_BENCH_TABLE=$((1<<29)); unset KMC_BENCH_FRONTIER
one config4_runtime "" --workload Kip279,5,2,2,1 --steps 2 --warmup 1
one config4_static "-DKMC_RT_GUARDS_MIN_INSTANCES=100000" --workload Kip279,5,2,2,1 --steps 2 --warmup 1
