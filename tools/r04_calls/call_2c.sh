#!/bin/bash
# round 4, call 2c: the wide kernel with THREE blocks per CU.  Call 2a's 3-5-wave variants never got their waves: a 10-word
# state needs 60 KB of LDS per block (ring 4 x 10 x 128 x 8 + stager 4 x 10 x 64 x 8), i.e. two blocks per CU whatever the
# registers allow.  With a 32-entry stager (KMC_QCAP_WIDE=32) it is 50 KB and three fit.  Ten levels of BASELINE config 5, counts
# checked against tests/golden/oracle_kip320_7_8_8_3_levels10.json; code objects prebuilt in kmc_cache_exp.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04_2c; mkdir -p $O
export KMC_NO_TORCH=1 KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29)) KMC_CACHE_DIR=$PWD/kafka_specification_amd/kmc_cache_exp
one() {  # tag, defines
  KMC_VERBOSE=1 KMC_JIT_DEFINES="$2" timeout 400 python bench.py --workload Kip320,7,8,8,3 --level-budget 10 \
     --no-cpu-baseline --no-orbit-counting --no-cold-start --steps 3 --warmup 1 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print("%-20s ms/step %7.2f kernel %7.2f ms  distinct %d generated %d depth %d %s" % (sys.argv[2], d["ms_per_step"], 1e3*d["roofline"]["kernel_seconds_per_step"], c["distinct_states"], c["states_generated"], c["depth"], c["verdict"]))
except Exception as e: print(sys.argv[2], "no line:", e)
PY
  grep -h "specialising" $O/$1.err | head -2
}
G="-DKMC_GROUPED_GUARDS_MIN_INSTANCES=100"; Q="-DKMC_QCAP_WIDE=32"
one static ""
one static_q32 "$Q"
one static_q32_w3 "$Q -DKMC_MIN_WAVES=3"
one grouped_w2 "$G -DKMC_MIN_WAVES=2"
one grouped_q32_w2 "$G $Q -DKMC_MIN_WAVES=2"
one grouped_q32_w3 "$G $Q -DKMC_MIN_WAVES=3"
