#!/bin/bash
# Runs bench.py on the headline workload for a list of kernel variants (JIT defines / launch
# knobs), one line per variant -> gpurun_out/sweep.log.  Usage: tools/sweep.sh "name|defines|blocks_per_cu" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export KMC_NO_TORCH=1
for spec in "$@"; do
  IFS='|' read -r name defs bpc <<< "$spec"
  export KMC_JIT_DEFINES="$defs"
  if [ -n "$bpc" ]; then export KMC_BLOCKS_PER_CU="$bpc"; else unset KMC_BLOCKS_PER_CU; fi
  out=$(timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1)
  echo "$name | $defs | bpc=$bpc | $(echo "$out" | python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); print('ms=%.1f Gstates/s=%.3f kernel_ms=%.1f ok=%s' % (r['ms_per_step'], r['value']/1e9, 1e3*r['roofline']['kernel_seconds_per_step'], r['config']['matches_oracle_golden']))
except Exception as e: print('FAILED', e)
")" | tee -a gpurun_out/sweep.log
done
