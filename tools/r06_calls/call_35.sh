#!/bin/bash
# round 6, call 35: which physical chunks are fast?  tools/membench/placement on a fresh box: 128 GiB mapped from 8 MiB chunks,
# every 1 GiB group measured (streaming memset, random stores, the seen-set's load + CAS mix); then the same again in a second
# process (does the picture repeat?), and the headline once before / between / after (where does ITS table land?).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/r06_calls/call_35.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_35; mkdir -p $O
( cd tools/membench && hipcc --offload-arch=gfx950 -O3 placement.hip -o placement 2>/dev/null ) || echo "build failed"
export KMC_NO_TORCH=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-traces-leg --no-cold-start --no-baseline-configs --no-stretch --steps 3 --warmup 1"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); b = j['config'].get('step_breakdown') or {}
        print('$1', 'k_expand %.2f ms, clear %.2f ms' % (b.get('k_expand_ms', 0), b.get('clear_seen_set_ms', 0)))
"; }
timeout 120 $B 2>/dev/null | pick "[headline, the box's first process]" | tee -a $O/placement.txt
timeout 300 tools/membench/placement 128 2>&1 | tee -a $O/placement.txt
timeout 120 $B 2>/dev/null | pick "[headline between]" | tee -a $O/placement.txt
timeout 300 tools/membench/placement 128 2>&1 | tee -a $O/placement.txt
timeout 120 $B 2>/dev/null | pick "[headline after]" | tee -a $O/placement.txt
timeout 300 tools/membench/placement 250 2>&1 | tee -a $O/placement.txt
