#!/bin/bash
# round 6, call 2: (1) k_inv in its streaming form (no guard predicates, replica after replica) at 4 / 3 / 2 waves per SIMD,
# (2) config 5's k_expand with the deferred probe's load no longer forced to wait, (3) probe chains walked 1 / 2 / 4 / 8 slots per round
# trip: the stretch (wide entries), the headline, config 5, (4) config 4 at SURVEY's own sizing over twelve levels, (5) the new GPU tests
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_2; mkdir -p $O
export KMC_NO_TORCH=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-cold-start --no-baseline-configs --no-stretch"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']
        print('$1', 'ms/step %.2f' % j['ms_per_step'], {k: round(v, 3) for k, v in (c.get('step_breakdown') or {}).items()}, 'golden', c['matches_oracle_golden'], 'frac %.4f' % j['roofline']['frac'])
"; }
C5="--workload Kip320,7,8,8,3 --level-budget 10 --steps 3 --warmup 1"
export KMC_BENCH_FRONTIER=$((1<<29))
echo "== 1+2. config 5 (table 2^31): k_inv variants, k_expand" | tee $O/config5.txt
for d in "" "-DKMC_INV_WAVES=3" "-DKMC_INV_WAVES=2" "-DKMC_PROBE_AHEAD=1" ""; do
  KMC_JIT_DEFINES="$d" KMC_BENCH_TABLE=$((1<<31)) timeout 300 $B $C5 2>>$O/err.txt | pick "[$d]" | tee -a $O/config5.txt
done
unset KMC_BENCH_FRONTIER
echo "== 3. probe chains: slots per round trip from the second step on" | tee $O/ahead.txt
for d in "-DKMC_PROBE_AHEAD=1" "" "-DKMC_PROBE_AHEAD=2" "-DKMC_PROBE_AHEAD=8" "-DKMC_PROBE_AHEAD=1" ""; do
  KMC_JIT_DEFINES="$d" timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline $d]" | tee -a $O/ahead.txt
done
for d in "-DKMC_PROBE_AHEAD=1" "" "-DKMC_PROBE_AHEAD=2" "-DKMC_PROBE_AHEAD=8"; do
  echo "[stretch wide $d]" | tee -a $O/ahead.txt
  tag=$(echo "$d" | tr -dc 0-9); tag=${tag:-4}
  KMC_JIT_DEFINES="$d" KMC_STRETCH_RUNS=2 KMC_STRETCH_LEVELS=$O/stretch_levels_wide_ahead$tag.jsonl timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | cut -c1-420 | tee -a $O/ahead.txt
done
for d in "-DKMC_PROBE_AHEAD=1" ""; do
  echo "[stretch narrow 2^33 $d]" | tee -a $O/ahead.txt
  KMC_NARROW=1 KMC_JIT_DEFINES="$d" KMC_STRETCH_RUNS=2 timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | cut -c1-420 | tee -a $O/ahead.txt
done
echo "[stretch narrow 2^34]" | tee -a $O/ahead.txt
KMC_NARROW=1 KMC_STRETCH_TABLE_LOG2=34 KMC_STRETCH_RUNS=2 KMC_STRETCH_LEVELS=$O/stretch_levels_narrow34.jsonl timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | cut -c1-420 | tee -a $O/ahead.txt
echo "== 4. config 4 at SURVEY's sizing, twelve levels" | tee $O/config4_deep.txt
KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<28)) timeout 300 $B --workload Kip279,5,4,4,3 --level-budget 12 --steps 3 --warmup 1 2>>$O/err.txt | pick "[Kip279 5/4/4/3, 12 levels]" | tee -a $O/config4_deep.txt
echo "== 5. tests"
unset KMC_NO_TORCH
timeout 1500 python -m pytest tests/test_gpu_insert_race.py tests/test_gpu_zz_beyond_the_exact_oracle.py tests/test_gpu_oracle_r_successors.py tests/test_gpu_selfcheck_and_fp128.py tests/test_gpu_deferred_probe.py -m gpu -x -q -k "multiset or config4 or 5-4-4-3 or fp128 or wide or selfcheck or collision or deferred" 2>&1 | tail -15 | tee $O/tests.txt
tail -20 $O/err.txt
