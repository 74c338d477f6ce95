#!/bin/bash
# round 6, call 4: with the seen-set any multiple of 64 slots, where is each leg's best table size?  (k_expand falls with the load,
# the clear at the start of a step grows with the slots: one box, every size twice, interleaved)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_4; mkdir -p $O
export KMC_NO_TORCH=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-cold-start --no-baseline-configs --no-stretch"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; b = c.get('step_breakdown') or {}
        print('$1', 'ms/step %.2f' % j['ms_per_step'], 'k_expand %.2f k_inv %.2f clear %.2f' % (b.get('k_expand_ms', 0), b.get('k_inv_ms', 0), b.get('clear_seen_set_ms', 0)), 'golden', c['matches_oracle_golden'], 'frac %.4f' % j['roofline']['frac'])
"; }
G=$((1<<30))
for rep in 1 2; do
  for m in 1000 1250 1500 1750 2000 2500; do
    KMC_BENCH_TABLE=$((G/1000*m)) timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline, $m/1000 x 2^30]" | tee -a $O/headline.txt
  done
  for m in 1000 1500 2000 3000; do
    KMC_BENCH_TABLE=$((G/4000*m)) timeout 300 $B --symmetry --steps 5 --warmup 1 2>>$O/err.txt | pick "[orbit counting, $m/1000 x 2^28]" | tee -a $O/sym.txt
  done
  for m in 1000 1500 2000; do
    KMC_BENCH_TABLE=$((G/1000*m)) timeout 300 $B --workload Kip279,5,2,2,1 --steps 5 --warmup 1 2>>$O/err.txt | pick "[config4, $m/1000 x 2^30]" | tee -a $O/config4.txt
  done
  for m in 1250 1500 1750 2000; do
    KMC_BENCH_TABLE=$((G/1000*m)) KMC_BENCH_FRONTIER=$((1<<29)) timeout 300 $B --workload Kip320,7,8,8,3 --level-budget 10 --steps 3 --warmup 1 2>>$O/err.txt | pick "[config5, $m/1000 x 2^30]" | tee -a $O/config5.txt
  done
  for m in 1500 2000 2500 3000; do
    KMC_BENCH_TABLE=$((G/1000*m)) KMC_BENCH_FRONTIER=$((1<<28)) timeout 300 $B --workload Kip279,5,4,4,3 --level-budget 12 --steps 3 --warmup 1 2>>$O/err.txt | pick "[config4 deep, $m/1000 x 2^30]" | tee -a $O/config4_deep.txt
  done
done
for t in 13.5e9 14.5e9 15.5e9; do
  echo "[stretch wide, $t slots, frontier 6.0e8]" | tee -a $O/stretch.txt
  KMC_STRETCH_TABLE=$t KMC_STRETCH_FRONTIER=6.0e8 KMC_STRETCH_RUNS=2 timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | cut -c1-460 | tee -a $O/stretch.txt
done
tail -5 $O/err.txt
