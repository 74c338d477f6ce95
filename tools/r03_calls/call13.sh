#!/bin/bash
# round 3, GPU call 13: why is the kind-major walk slower with the table on (41.9 ms against 35.0) while faster with it off
# (21.7 against 25.9)?  Atomic and read requests of the two builds (rocprofv3 --pmc, separate passes)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/c13; mkdir -p $O
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
export KMC_NO_TORCH=1
CMD="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
for v in km im; do
  if [ $v = km ]; then export KMC_JIT_DEFINES="-DKMC_MIN_WAVES=6"; unset KMC_LAYOUT; else unset KMC_JIT_DEFINES; export KMC_LAYOUT=tight; fi
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $set | cut -d' ' -f1)
    timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${v}_$n -o pmc -- $CMD > $O/${v}_$n.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
for v in ("km", "im"):
    tot = collections.defaultdict(float)
    for f in glob.glob("$O/%s_*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("kmc_expand"):
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    print(v, {k: "%.4g" % x for k, x in sorted(tot.items())})
PY
