#!/bin/bash
# round 6, call 31: calls 29 and 30 read the default bench line's headline at 31.7 ms of k_expand with a 1.34 - 1.36 ms clear - the
# signature of a seen-set in one hipMalloc - where call 28 read 28.7 / 1.22 on the same code, and where the same search without
# torch in the process (KMC_NO_TORCH=1: the system's HIP runtime instead of the one torch bundles) ran at 28.4 in every process
# of call 30.  Which memory does a process WITH torch get (KMC_VERBOSE says, and why a mapping failed)?  Interleaved, four times.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/r06_calls/call_31.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_31; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-traces-leg --no-cold-start --no-baseline-configs --no-stretch --steps 5 --warmup 1"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; b = c.get('step_breakdown') or {}
        print('$1', 'ms/step %.2f' % j['ms_per_step'], 'k_expand %.2f clear %.2f' % (b.get('k_expand_ms', 0), b.get('clear_seen_set_ms', 0)), 'golden', c['matches_oracle_golden'])
"; }
for rep in 1 2 3 4; do
  KMC_VERBOSE=1 timeout 300 $B 2>$O/e.txt | pick "[rep $rep: torch in the process]" | tee -a $O/torch.txt; grep -E "seen-set|released" $O/e.txt | cut -c1-300 | tee -a $O/torch.txt
  KMC_NO_TORCH=1 KMC_VERBOSE=1 timeout 300 $B 2>$O/e.txt | pick "[rep $rep: no torch]" | tee -a $O/torch.txt; grep -E "seen-set|released" $O/e.txt | cut -c1-300 | tee -a $O/torch.txt
done
python - <<'PY' 2>&1 | tee -a $O/torch.txt
import ctypes, os
import torch
print("torch", torch.__version__, "hip", torch.version.hip)
for l in open("/proc/self/maps"):
    if "libamdhip64" in l or "libhsa-runtime" in l:
        print(l.split()[-1]); 
PY
