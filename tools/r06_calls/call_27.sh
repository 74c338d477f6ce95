#!/bin/bash
# round 6, call 27: a run that keeps traces on 64-bit entries stores a claim's predecessor in the claim's own slot (16-byte slots,
# KMC_FLAG_PAIRED) against the predecessor table of its own (KMC_PAIRED_SLOTS=0): the headline, BASELINE configs 4 and 5 with
# KMC_BENCH_TRACE=1, fresh processes, interleaved, counts exact; the same legs without traces (the claim loop is now a template
# over the slot's stride: nothing may have moved); then the tests that touch traces, checkpoints and the seen-set's memory.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r06_calls/call_27.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_27; mkdir -p $O
export KMC_NO_TORCH=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-cold-start --no-baseline-configs --no-stretch"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; b = c.get('step_breakdown') or {}
        print('$1', 'ms/step %.2f' % j['ms_per_step'], 'k_expand %.2f k_inv %.2f clear %.2f' % (b.get('k_expand_ms', 0), b.get('k_inv_ms', 0), b.get('clear_seen_set_ms', 0)), 'golden', c['matches_oracle_golden'])
"; }
for rep in 1 2 3; do
  for tr in "no traces" "paired" "separate"; do
    unset KMC_BENCH_TRACE KMC_PAIRED_SLOTS
    [ "$tr" != "no traces" ] && export KMC_BENCH_TRACE=1
    [ "$tr" = "separate" ] && export KMC_PAIRED_SLOTS=0
    timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline, $tr]" | tee -a $O/ab.txt
    timeout 300 $B --workload Kip279,5,2,2,1 --steps 5 --warmup 1 2>>$O/err.txt | pick "[config4, $tr]" | tee -a $O/ab.txt
    KMC_BENCH_TABLE=$((7<<28)) KMC_BENCH_FRONTIER=$((1<<29)) timeout 300 $B --workload Kip320,7,8,8,3 --level-budget 10 --steps 3 --warmup 1 2>>$O/err.txt | pick "[config5, $tr]" | tee -a $O/ab.txt
  done
done
unset KMC_BENCH_TRACE KMC_PAIRED_SLOTS KMC_NO_TORCH
( time timeout 1500 python -m pytest tests/test_gpu_seen_set_memory.py tests/test_gpu_sharded_and_traces.py tests/test_gpu_selfcheck_and_fp128.py tests/test_gpu_deferred_probe.py tests/test_gpu_insert_race.py -x -q -m gpu ) > $O/pytest_some.log 2>&1; tail -15 $O/pytest_some.log
tail -5 $O/err.txt
