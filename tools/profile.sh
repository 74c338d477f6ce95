#!/bin/bash
# rocprofv3 passes over the headline bench (system ROCm, no torch in the process):
#   1. --kernel-trace --stats                  -> per-kernel durations
#   2..n. --pmc <set> (one set per pass)       -> HBM bytes, L2 hit rate, SQ stall breakdown
# Output: gpurun_out/prof_<tag>/...csv ; summarise with tools/summarize_profile.py
TAG=${1:-r02}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export KMC_NO_TORCH=1
# PROFILE_BENCH_ARGS: e.g. --symmetry (the orbit-counting search in the timed region); the default line's orbit_counting leg is
# left out of a profiled run either way, so that every kmc_expand launch in the trace belongs to ONE search
CMD="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-orbit-counting --no-traces-leg --no-cold-start --no-baseline-configs --no-stretch ${PROFILE_BENCH_ARGS:-}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1
i=0
# (the two *_DRAM_32B passes are the HBM byte count: on gfx950 FETCH_SIZE tallies a 128-byte read request as 64 bytes and
#  every random probe of this kernel is one — profiles/r02_request_size.txt)
for set in "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout ${PROFILE_PMC_TIMEOUT:-90} rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -o pmc -- $CMD > "$OUT/pmc$i.log" 2>&1
  echo "pass $i ($set): rc=$?" >> "$OUT/passes.log"
done
# roctx ranges (one per BFS level, or per chain of levels when kmc_run chains them; the clear of the seen-set and
# the invariant pass have their own): a marker trace WITHOUT counters (gpurun refuses --pmc together with a marker trace)
KMC_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace --output-format csv -d "$OUT/markers" -o markers -- $CMD > "$OUT/markers.log" 2>&1
MK="$(find "$OUT/markers" -name '*marker_api_trace.csv' | head -1)"
[ -n "$MK" ] && { echo "roctx ranges in the trace: $(grep -c 'kmc' "$MK")"; grep 'kmc' "$MK" | head -4 | cut -c1-220; cp "$MK" "$OUT/marker_api_trace.csv"; }
grep -h '"metric"' "$OUT"/*.log | head -1 | cut -c1-400
find "$OUT" -name "*.csv" | head -30
python3 "$REPO/tools/summarize_profile.py" "$OUT" "$OUT/summary.json" "$OUT/pmc_summary.json" > /dev/null && echo summarised
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
