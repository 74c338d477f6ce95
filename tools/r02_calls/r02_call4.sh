#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c4
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/gpu_tests.txt 2>&1
echo "pytest rc=$?" >> $OUT/gpu_tests.txt
tail -40 $OUT/gpu_tests.txt | cut -c1-250
export KMC_NO_TORCH=1
for t in 1073741824 536870912; do
  KMC_BENCH_TABLE=$t timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); print('table $t ms=%.2f kernel_ms=%.2f ok=%s' % (r['ms_per_step'], 1e3*r['roofline']['kernel_seconds_per_step'], r['config']['matches_oracle_golden']))" >> $OUT/table_ab.txt
done
cat $OUT/table_ab.txt
nproc > $OUT/oracle_scaling.txt
for T in 8 16 32 64 128 256; do
KMO_TIMING=1 timeout 120 python - $T >> $OUT/oracle_scaling.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import kmo
T = int(sys.argv[1])
cfg = kmo.make_config("Kip320", N=3, L=6, R=6, E=2, invariants=("TypeOk","WeakIsr","StrongIsr"), threads=T, max_states=4_000_000)
r = kmo.Run(cfg)
print(T, r.distinct, r.depth, "%.2fs" % r.seconds, "%.2f M/s" % (r.distinct/r.seconds/1e6), flush=True)
PY
done
cat $OUT/oracle_scaling.txt
bash tools/r02_call3c.sh
