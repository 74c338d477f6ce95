#!/bin/bash
# round 4, call 2a: the wide kernel (BASELINE config 5, ten levels) — phase split and register-budget / guard-group variants,
# every run checked against tests/golden/oracle_kip320_7_8_8_3_levels10.json.  Code objects come prebuilt in kmc_cache_exp.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r04_calls/call_2a.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04_2a; mkdir -p $O
export KMC_NO_TORCH=1 KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29)) KMC_CACHE_DIR=$PWD/kafka_specification_amd/kmc_cache_exp
one() {  # tag, defines, extra bench args
  KMC_JIT_DEFINES="$2" timeout 400 python bench.py --workload ${WL:-Kip320,7,8,8,3} ${BUDGET---level-budget 10} $3 \
     --no-cpu-baseline --no-orbit-counting --no-cold-start --steps 3 --warmup 1 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print("%-16s ms/step %7.2f kernel %7.2f ms  distinct %d generated %d depth %d %s" % (sys.argv[2], d["ms_per_step"], 1e3*d["roofline"]["kernel_seconds_per_step"], c["distinct_states"], c["states_generated"], c["depth"], c["verdict"]))
except Exception as e: print(sys.argv[2], "no line:", e)
PY
  grep -h "\[kmc\]" $O/$1.err | sort -u | tail -3
}
G="-DKMC_GROUPED_GUARDS_MIN_INSTANCES=100"
one static ""
one static_w3 "-DKMC_MIN_WAVES=3"
one static_prof "-DKMC_PROFILE=1"
one grouped "$G"
one grouped_w3 "$G -DKMC_MIN_WAVES=3"
one grouped_w4 "$G -DKMC_MIN_WAVES=4"
one grouped_w5 "$G -DKMC_MIN_WAVES=5"
one grouped_w4_prof "$G -DKMC_MIN_WAVES=4 -DKMC_PROFILE=1"
# orbit counting on the same configuration, 14 levels (golden: 50,390,682,994 states)
BUDGET="--level-budget 14"
one sym_static "" --symmetry
one sym_grouped_w2 "$G -DKMC_MIN_WAVES=2" --symmetry
one sym_grouped_w3 "$G -DKMC_MIN_WAVES=3" --symmetry
# BASELINE config 4 (exhaustible: 112,549,196 states)
unset KMC_BENCH_TABLE KMC_BENCH_FRONTIER
WL=Kip279,5,2,2,1 BUDGET=""
one c4_static ""
one c4_grouped "$G"
