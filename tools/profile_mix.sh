#!/bin/bash
# instruction-mix PMC pass.  usage: tools/profile_mix.sh <tag>
TAG=${1:-mix}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp KMC_NO_TORCH=1
CMD="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc1" -o pmc -- $CMD > "$OUT/pmc1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM --output-format csv -d "$OUT/pmc2" -o pmc -- $CMD > "$OUT/pmc2.log" 2>&1
python $REPO/tools/summarize_profile.py "$OUT" 2>/dev/null | python -c "
import sys,json; r=json.load(sys.stdin); print(json.dumps({'kernel_ms':1e3*r['kernel_seconds_total'],'ctr':r['counters_sum_over_launches']}))"
