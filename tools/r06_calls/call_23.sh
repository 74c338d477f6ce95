#!/bin/bash
# (KMC_TABLE_VMM / KMC_FRONTIER_VMM were the hooks of this experiment in kmc_open; what came of it is KmcEngine's seen_set_alloc and
# KMC_SEEN_SET_CHUNK_LOG2 - csrc/kmc_engine_core.cpp - with which these A/Bs are: chunks = the default, hipMalloc = KMC_SEEN_SET_CHUNK_LOG2=0)
# round 6, call 23: the seen-set (and the frontiers) mapped from 8 MiB chunks against hipMalloc on every leg of the bench line, fresh
# processes, interleaved, counts against the exact fixtures in every run
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_23; mkdir -p $O
export KMC_NO_TORCH=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-cold-start --no-baseline-configs --no-stretch"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; b = c.get('step_breakdown') or {}
        print('$1', 'ms/step %.2f' % j['ms_per_step'], 'k_expand %.2f k_inv %.2f clear %.2f' % (b.get('k_expand_ms', 0), b.get('k_inv_ms', 0), b.get('clear_seen_set_ms', 0)), 'golden', c['matches_oracle_golden'])
"; }
for rep in 1 2 3; do for v in "none" "table" "both"; do
  unset KMC_TABLE_VMM KMC_FRONTIER_VMM
  [ $v != none ] && export KMC_TABLE_VMM=23
  [ $v = both ] && export KMC_FRONTIER_VMM=23
  timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline $v]" | tee -a $O/ab.txt
  timeout 300 $B --symmetry --steps 5 --warmup 1 2>>$O/err.txt | pick "[orbit counting $v]" | tee -a $O/ab.txt
  timeout 300 $B --workload Kip279,5,2,2,1 --steps 5 --warmup 1 2>>$O/err.txt | pick "[config4 $v]" | tee -a $O/ab.txt
  KMC_BENCH_TABLE=$((7<<28)) KMC_BENCH_FRONTIER=$((1<<29)) timeout 300 $B --workload Kip320,7,8,8,3 --level-budget 10 --steps 3 --warmup 1 2>>$O/err.txt | pick "[config5 $v]" | tee -a $O/ab.txt
  KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<28)) timeout 300 $B --workload Kip279,5,4,4,3 --level-budget 12 --steps 3 --warmup 1 2>>$O/err.txt | pick "[config4 deep $v]" | tee -a $O/ab.txt
done; done
for v in none table none table; do
  unset KMC_TABLE_VMM; [ $v != none ] && export KMC_TABLE_VMM=23
  echo "[stretch wide, 15e9 slots, $v]" | tee -a $O/stretch.txt
  KMC_STRETCH_TABLE=15e9 KMC_STRETCH_FRONTIER=6.0e8 KMC_STRETCH_RUNS=2 timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | cut -c1-330 | tee -a $O/stretch.txt
done
tail -5 $O/err.txt
