#!/bin/bash
# request size of a random probe: timing of every load flavour / allocation type, then the gfx950 request-size counters
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/r02c16
mkdir -p $OUT
BIN=$REPO/tools/membench/reqsize
timeout 120 $BIN 30 200 > $OUT/timing.txt 2>&1
cat $OUT/timing.txt
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_BUBBLE_sum" "TCC_READ_SECTORS_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  i=$((i+1))
  timeout 60 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc$i -o pmc -- $BIN 30 50 > $OUT/pmc$i.log 2>&1
  echo "pass $i ($ctr) rc=$?" >> $OUT/passes.log
done
cat $OUT/passes.log
OUT=$OUT python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "/root/repo/gpurun_out/r02c16")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc*/**/pmc_counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    # two dispatches per kernel name (warm-up of 4 iterations, then the measured 50): keep the larger dispatch id
    last = {}
    for r in rows:
        last[r["Kernel_Name"]] = max(last.get(r["Kernel_Name"], 0), int(r["Dispatch_Id"]))
    for r in rows:
        if int(r["Dispatch_Id"]) == last[r["Kernel_Name"]]:
            acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
n = 2048 * 256 * 50
with open(out + "/requests_per_lane_iteration.txt", "w") as fo:
    for kname in sorted(acc):
        line = kname.split("(")[0] + "  " + "  ".join(f"{c}={v / n:.3f}" for c, v in sorted(acc[kname].items()))
        print(line); fo.write(line + "\n")
PY
