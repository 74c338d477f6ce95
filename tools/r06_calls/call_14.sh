#!/bin/bash
# round 6, call 14: call 13 showed that the headline's mode is carried mostly by WHERE THE SEEN-SET LIES.  Is it address translation?
# One handle whose table is moved before every search (KMC_DEBUG_REALLOC=table), under rocprofv3 with the translation counters:
# per search, k_expand's time against UTCL2 busy cycles / UTCL1 misses / stalls on UTCL2 credits.
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_14; mkdir -p $O
export KMC_NO_TORCH=1
cat > /tmp/moves.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import kafka_specification_amd as kmc
from kafka_specification_amd.configs import HEADLINE
with kmc.ModelChecker(kmc.CheckerConfig(**HEADLINE, table_capacity=3 << 29, frontier_capacity=1 << 26)) as mc:
    for k in range(12):
        r = mc.run()
        print("search %d k_expand %.2f ms" % (k, r.seconds_expand * 1e3), flush=True)
PY
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"; do
  i=$((i+1))
  KMC_DEBUG_REALLOC=table timeout 280 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -o pmc -- python /tmp/moves.py > $O/pmc$i.log 2>&1
  grep "^search" $O/pmc$i.log | tr '\n' ';'; echo
  python - $O/pmc$i <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not cc or not kt:
    print("no csv", cc, kt); sys.exit()
dur = {}
order = []
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    order.append((int(r["Start_Timestamp"]), r["Dispatch_Id"]))
order.sort()
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(cc[0])):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = cnt[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
search, rows = -1, []
for _, did in order:
    name, ns = dur[did]
    if name.startswith("kmc_insert"):
        search += 1; rows.append([0, collections.defaultdict(float)])
    elif name.startswith("kmc_expand") and search >= 0:
        rows[search][0] += ns
        for k, v in cnt.get(did, {}).items():
            rows[search][1][k] += v
for k, (ns, c) in enumerate(rows):
    print("  search %2d k_expand %.2f ms (under the profiler) " % (k, ns / 1e6) + "  ".join("%s %.4g" % (n, v) for n, v in sorted(c.items())))
PY
done 2>&1 | tee $O/translation.txt
