#!/bin/bash
# round 6, call 9: a cheaper fingerprint for wide states?  (-DKMC_FOLD_MIN_WORDS=8: two words absorbed per 64 x 64 -> 128 multiply,
# high half xor low half, one full-avalanche finaliser at the end - a third of the multiplies of the per-word mix64 chain.)
# BASELINE config 5, same box, interleaved, counts against the exact ten-level fixture in every run.
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_9; mkdir -p $O
export KMC_NO_TORCH=1 KMC_BENCH_TABLE=$(( (7<<30)/4 )) KMC_BENCH_FRONTIER=$((1<<29))
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-cold-start --no-baseline-configs --no-stretch"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; b = c.get('step_breakdown') or {}
        print('$1', 'ms/step %.2f' % j['ms_per_step'], 'k_expand %.2f k_inv %.2f clear %.2f' % (b.get('k_expand_ms', 0), b.get('k_inv_ms', 0), b.get('clear_seen_set_ms', 0)), 'golden', c['matches_oracle_golden'], 'frac %.4f' % j['roofline']['frac'])
"; }
for rep in 1 2 3; do for d in "" "-DKMC_FOLD_MIN_WORDS=8"; do
  KMC_JIT_DEFINES="$d" timeout 300 $B --workload Kip320,7,8,8,3 --level-budget 10 --steps 3 --warmup 1 2>>$O/err.txt | pick "[config5 $d]" | tee -a $O/ab.txt
done; done
tail -5 $O/err.txt
