#!/bin/bash
# round 6, call 1: (1) the k_inv sweep on BASELINE config 5 (waves per SIMD x prefetch), (2) the seen-set's size on that leg,
# (3) the two compilers A/B on the four profiled kernels, (4) randbench from 8 GiB to 128 GiB, (5) the 6.45 G-state stretch with
# per-level probe rates, wide and narrow, (6) a rocprofv3 profile of config 5 with the k_inv kernel in it
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_1; mkdir -p $O
export KMC_NO_TORCH=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-cold-start --no-baseline-configs --no-stretch"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']
        print('$1', 'ms/step %.2f' % j['ms_per_step'], {k: round(v, 3) for k, v in (c.get('step_breakdown') or {}).items()}, 'golden', c['matches_oracle_golden'], 'compiler', (j['roofline'].get('kernel_compiler') or {}).get('hip_runtime_version_of_the_compiler'))
"; }
C5="--workload Kip320,7,8,8,3 --level-budget 10 --steps 3 --warmup 1"
echo "== 1. k_inv variants (config 5, table 2^31)" | tee $O/inv.txt
for d in "" "-DKMC_INV_WAVES=2" "-DKMC_INV_WAVES=3" "-DKMC_INV_WAVES=4 -DKMC_INV_PREFETCH=0" "-DKMC_INV_WAVES=6" "-DKMC_INV_WAVES=8" "-DKMC_INV_WAVES=2 -DKMC_INV_PREFETCH=0"; do
  KMC_JIT_DEFINES="$d" KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29)) timeout 300 $B $C5 2>>$O/err.txt | pick "[$d]" | tee -a $O/inv.txt
done
echo "== 2. table size (config 5, default k_inv)" | tee $O/table.txt
for t in 31 30 29 31 30 29; do
  KMC_BENCH_TABLE=$((1<<t)) KMC_BENCH_FRONTIER=$((1<<29)) timeout 300 $B $C5 2>>$O/err.txt | pick "[2^$t]" | tee -a $O/table.txt
done
echo "== 3. compilers (70051831 = PyTorch's bundle, 70226015 = system ROCm 7.2)" | tee $O/compilers.txt
for rep in 1 2; do for pin in 70051831 70226015; do
  export KMC_COMPILER_PIN=$pin
  timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline pin $pin]" | tee -a $O/compilers.txt
  timeout 300 $B --symmetry --steps 5 --warmup 1 2>>$O/err.txt | pick "[orbit counting pin $pin]" | tee -a $O/compilers.txt
  timeout 300 $B --workload Kip279,5,2,2,1 --steps 5 --warmup 1 2>>$O/err.txt | pick "[config4 pin $pin]" | tee -a $O/compilers.txt
  KMC_BENCH_TABLE=$((1<<30)) KMC_BENCH_FRONTIER=$((1<<29)) timeout 300 $B $C5 2>>$O/err.txt | pick "[config5 pin $pin]" | tee -a $O/compilers.txt
done; done
unset KMC_COMPILER_PIN
echo "== 4. randbench footprints" | tee $O/randbench_sweep.txt
RANDBENCH_MAX_LOG2=34 RANDBENCH_MODES=1,3,7,13 timeout 600 tools/membench/randbench 0 27 30 31 32 33 34 2>&1 | tee -a $O/randbench_sweep.txt
echo "-- table at a 1 GiB-aligned address" | tee -a $O/randbench_sweep.txt
RANDBENCH_ALIGN_GIB=1 RANDBENCH_MAX_LOG2=34 RANDBENCH_MODES=1,7,13 timeout 600 tools/membench/randbench 0 30 33 34 2>&1 | tee -a $O/randbench_sweep.txt
echo "== 5. stretch" | tee $O/stretch.txt
KMC_STRETCH_RUNS=2 KMC_STRETCH_LEVELS=$O/stretch_levels_wide33.jsonl timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | tee -a $O/stretch.txt
KMC_NARROW=1 KMC_STRETCH_RUNS=2 KMC_STRETCH_LEVELS=$O/stretch_levels_narrow33.jsonl timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | tee -a $O/stretch.txt
KMC_STRETCH_TABLE_LOG2=32 KMC_STRETCH_RUNS=2 KMC_STRETCH_LEVELS=$O/stretch_levels_wide32.jsonl timeout 600 python tools/fp128_stretch.py 0 2>>$O/err.txt | tee -a $O/stretch.txt
echo "== 6. profile of config 5 (k_expand + k_inv)"
KMC_BENCH_TABLE=$((1<<30)) KMC_BENCH_FRONTIER=$((1<<29)) PROFILE_BENCH_ARGS="--no-stretch --workload Kip320,7,8,8,3 --level-budget 10" timeout 900 tools/profile.sh r06_config5 2>&1 | tail -5
tail -20 $O/err.txt
