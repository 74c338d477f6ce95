#!/bin/bash
# round 6, call 3: the seen-set of any multiple of 64 slots (kmc_slot_of): (1) nothing lost on the four profiled workloads, (2) the
# stretch with a table sized to the HBM, (3) k_inv with the next tile prefetched (83 registers now), (4) the whole default bench line,
# (5) the parity suite on the new index
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_3; mkdir -p $O
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
echo "== 1. the four workloads on the new slot function" | tee $O/workloads.txt
for rep in 1 2; do
  timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline]" | tee -a $O/workloads.txt
  timeout 300 $B --symmetry --steps 5 --warmup 1 2>>$O/err.txt | pick "[orbit counting]" | tee -a $O/workloads.txt
  timeout 300 $B --workload Kip279,5,2,2,1 --steps 5 --warmup 1 2>>$O/err.txt | pick "[config4]" | tee -a $O/workloads.txt
  KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29)) timeout 300 $B $C5 2>>$O/err.txt | pick "[config5]" | tee -a $O/workloads.txt
  KMC_JIT_DEFINES="-DKMC_INV_PREFETCH=1" KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29)) timeout 300 $B $C5 2>>$O/err.txt | pick "[config5, k_inv prefetch]" | tee -a $O/workloads.txt
done
KMC_BENCH_TABLE=$((3<<29)) timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline, table 1.5 x 2^30]" | tee -a $O/workloads.txt
KMC_BENCH_TABLE=$((3<<30)) KMC_BENCH_FRONTIER=$((1<<29)) timeout 300 $B $C5 2>>$O/err.txt | pick "[config5, table 1.5 x 2^31]" | tee -a $O/workloads.txt
echo "== 2. stretch, wide entries, table sized to the HBM" | tee $O/stretch.txt
python -c "
import ctypes; h = ctypes.CDLL('libamdhip64.so'); f = ctypes.c_size_t(); t = ctypes.c_size_t(); h.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)); print('hipMemGetInfo free %.1f GB of %.1f GB' % (f.value / 1e9, t.value / 1e9))" | tee -a $O/stretch.txt
for t in 8589934592 11.0e9 12.0e9 12.9e9 13.5e9; do
  echo "[wide, $t slots, frontier 6.0e8]" | tee -a $O/stretch.txt
  KMC_STRETCH_TABLE=$t KMC_STRETCH_FRONTIER=6.0e8 KMC_STRETCH_RUNS=2 KMC_STRETCH_LEVELS=$O/stretch_levels_wide_$t.jsonl timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | cut -c1-460 | tee -a $O/stretch.txt
done
echo "[narrow, 2^34 slots]" | tee -a $O/stretch.txt
KMC_NARROW=1 KMC_STRETCH_TABLE_LOG2=34 KMC_STRETCH_FRONTIER=6.0e8 KMC_STRETCH_RUNS=2 timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | cut -c1-460 | tee -a $O/stretch.txt
echo "[narrow, 2.4e10 slots]" | tee -a $O/stretch.txt
KMC_NARROW=1 KMC_STRETCH_TABLE=2.4e10 KMC_STRETCH_FRONTIER=6.0e8 KMC_STRETCH_RUNS=2 timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | cut -c1-460 | tee -a $O/stretch.txt
echo "== 4. the default bench line (as the driver runs it)"
unset KMC_NO_TORCH
(time KMC_BENCH_STRETCH_TABLE=12900000000 KMC_BENCH_STRETCH_FRONTIER=600000000 timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err) 2>&1 | tail -3
python -c "
import json
j = json.load(open('$O/bench.json'))
print('headline ms', j['ms_per_step'], j['config']['step_breakdown'], j['config']['matches_oracle_golden'])
for k, v in j.get('baseline_configs', {}).items():
    print(k, v.get('ms_per_step'), v.get('step_breakdown'), v.get('matches_oracle_golden'), (v.get('k_inv') or {}).get('achieved'), (v.get('cpu_baseline') or {}).get('value'), v.get('error'))
s = j.get('stretch_1gpu', {})
print('stretch', {k: s.get(k) for k in ('time_to_exhaustive_s', 'first_run_wall_s', 'open_s', 'matches_oracle_golden', 'table_slots', 'table_load_at_end', 'error')}, (s.get('roofline') or {}).get('frac'), (s.get('roofline') or {}).get('probes_per_s'))
print('cpu', j.get('cpu_baseline', {}).get('value'), 'cold', (j.get('cold_start') or {}).get('wall_s'))
" 2>&1 | tee $O/bench_digest.txt
tail -5 $O/bench.err
echo "== 5. parity suite (what is precompiled or small)"
timeout 1500 python -m pytest tests/test_gpu_insert_race.py tests/test_gpu_parity.py tests/test_gpu_selfcheck_and_fp128.py tests/test_gpu_sharded_and_traces.py -m gpu -x -q 2>&1 | tail -8 | tee $O/tests.txt
tail -12 $O/err.txt
