#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c5
mkdir -p $OUT
nproc > $OUT/oracle_scaling.txt
for T in 16 32 64 128 256; do
KMO_TIMING=1 timeout 100 python - $T >> $OUT/oracle_scaling.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import kmo
T = int(sys.argv[1])
cfg = kmo.make_config("Kip320", N=3, L=6, R=6, E=2, invariants=("TypeOk","WeakIsr","StrongIsr"), threads=T, max_states=28_000_000)
r = kmo.Run(cfg)
print(T, r.distinct, r.depth, "%.2fs" % r.seconds, "%.2f M/s" % (r.distinct/r.seconds/1e6), flush=True)
PY
done
cat $OUT/oracle_scaling.txt
