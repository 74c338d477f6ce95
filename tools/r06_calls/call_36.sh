#!/bin/bash
# round 6, call 36: is it the diversity of a table's physical chunks?  tools/membench/diversity as the box's FIRST GPU work (an
# unfragmented allocator: consecutive chunks are physically consecutive), a 128 GiB pool; then the headline; then diversity again.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/r06_calls/call_36.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_36; mkdir -p $O
( cd tools/membench && hipcc --offload-arch=gfx950 -O3 diversity.hip -o diversity 2>/dev/null ) || echo "build failed"
timeout 300 tools/membench/diversity 128 2>&1 | tee -a $O/diversity.txt
export KMC_NO_TORCH=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-traces-leg --no-cold-start --no-baseline-configs --no-stretch --steps 3 --warmup 1"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); b = j['config'].get('step_breakdown') or {}
        print('$1', 'k_expand %.2f ms, clear %.2f ms' % (b.get('k_expand_ms', 0), b.get('clear_seen_set_ms', 0)))
"; }
timeout 120 $B 2>/dev/null | pick "[headline]" | tee -a $O/diversity.txt
timeout 300 tools/membench/diversity 128 2>&1 | tee -a $O/diversity.txt
timeout 120 $B 2>/dev/null | pick "[headline]" | tee -a $O/diversity.txt
timeout 300 tools/membench/diversity 256 2>&1 | tee -a $O/diversity.txt
