#!/bin/bash
# Quick SQ-only PMC pass (+ dry-run timing) for kernel tuning.  usage: tools/profile_sq.sh <tag>
TAG=${1:-sq}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export KMC_NO_TORCH=1
CMD="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
KMC_DRYRUN=1 $CMD 2>&1 | grep "kmc\]" | tee "$OUT/dry.log"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$OUT/pmc1" -o pmc -- $CMD > "$OUT/pmc1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS --output-format csv -d "$OUT/pmc2" -o pmc -- $CMD > "$OUT/pmc2.log" 2>&1
python $REPO/tools/summarize_profile.py "$OUT" 2>/dev/null | python -c "
import sys,json; r=json.load(sys.stdin); print(json.dumps({'kernel_ms':1e3*r['kernel_seconds_total'],'cfg':r['dominant_launch_cfg'],'ctr':r['counters_sum_over_launches'],'der':r['derived']}))"
