#!/bin/bash
# round 6, call 7: probe chains that advance ONE ROUND TRIP PER STEP FOR EVERY LANE (KmcSink::claim_steps / claim_wide_steps,
# -DKMC_MERGED_STEPS=1) against the textbook loop (claim_from), same box, interleaved: the four profiled workloads, config 4 at
# SURVEY's sizing, the stretch (wide and narrow entries) with per-level probe rates; counts against the exact fixtures in every run
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_7; mkdir -p $O
export KMC_NO_TORCH=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-cold-start --no-baseline-configs --no-stretch"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; b = c.get('step_breakdown') or {}
        print('$1', 'ms/step %.2f' % j['ms_per_step'], 'k_expand %.2f k_inv %.2f clear %.2f' % (b.get('k_expand_ms', 0), b.get('k_inv_ms', 0), b.get('clear_seen_set_ms', 0)), 'golden', c['matches_oracle_golden'], 'frac %.4f' % j['roofline']['frac'])
"; }
M="-DKMC_MERGED_STEPS=1"
for rep in 1 2; do for d in "" "$M"; do
  export KMC_JIT_DEFINES="$d"
  timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline $d]" | tee -a $O/ab.txt
  timeout 300 $B --symmetry --steps 5 --warmup 1 2>>$O/err.txt | pick "[orbit counting $d]" | tee -a $O/ab.txt
  timeout 300 $B --workload Kip279,5,2,2,1 --steps 5 --warmup 1 2>>$O/err.txt | pick "[config4 $d]" | tee -a $O/ab.txt
  KMC_BENCH_TABLE=$(( (7<<30)/4 )) KMC_BENCH_FRONTIER=$((1<<29)) timeout 300 $B --workload Kip320,7,8,8,3 --level-budget 10 --steps 3 --warmup 1 2>>$O/err.txt | pick "[config5 $d]" | tee -a $O/ab.txt
done; done
for d in "" "$M"; do
  export KMC_JIT_DEFINES="$d"; tag=$([ -z "$d" ] && echo loop || echo steps)
  echo "[stretch wide, 13.5e9 slots $d]" | tee -a $O/stretch.txt
  KMC_STRETCH_TABLE=13.5e9 KMC_STRETCH_FRONTIER=6.0e8 KMC_STRETCH_RUNS=2 KMC_STRETCH_LEVELS=$O/stretch_levels_wide_$tag.jsonl timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | cut -c1-460 | tee -a $O/stretch.txt
  echo "[stretch wide, 2^33 slots $d]" | tee -a $O/stretch.txt
  KMC_STRETCH_TABLE_LOG2=33 KMC_STRETCH_FRONTIER=6.0e8 KMC_STRETCH_RUNS=2 KMC_STRETCH_LEVELS=$O/stretch_levels_wide33_$tag.jsonl timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | cut -c1-460 | tee -a $O/stretch.txt
  echo "[stretch narrow, 2^34 slots $d]" | tee -a $O/stretch.txt
  KMC_NARROW=1 KMC_STRETCH_TABLE_LOG2=34 KMC_STRETCH_FRONTIER=6.0e8 KMC_STRETCH_RUNS=2 KMC_STRETCH_LEVELS=$O/stretch_levels_narrow_$tag.jsonl timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | cut -c1-460 | tee -a $O/stretch.txt
done
echo "== parity on the merged walk (small configurations: exact level sets; the wide table with collisions on demand)"
KMC_JIT_DEFINES="$M" timeout 1200 python -m pytest tests/test_gpu_insert_race.py tests/test_gpu_parity.py tests/test_gpu_selfcheck_and_fp128.py tests/test_gpu_deferred_probe.py -m gpu -x -q 2>&1 | tail -8 | tee $O/tests.txt
tail -12 $O/err.txt
