#!/usr/bin/env python3
"""Reads tools/calibrate_fetch.sh's counter CSVs: per randbench mode, counter value per random 8-byte access.
randbench dispatches kernel `k` twice per mode (8 warm-up iterations, then the measured one), 8 modes, 2048 blocks x 256
lanes; iterations per lane: 400 for modes 0 and 6, 800 otherwise.  FETCH_SIZE / WRITE_SIZE are reported in KiB."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
MODES = ["0 dependent loads", "1 independent loads x4", "2 CAS 0->x", "3 load then CAS", "4 plain stores",
         "5 no-return atomicMax", "6 sc1 loads", "7 load + 35 % CAS"]
ACC = [2048 * 256 * (400 if m in (0, 6) else 800) for m in range(8)]
vals = defaultdict(dict)   # counter -> {dispatch order index -> value}
for f in glob.glob(os.path.join(d, "*", "**", "pmc_counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k(") or r["Kernel_Name"] == "k"]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    order = {i: n for n, i in enumerate(ids)}
    for r in rows:
        c = vals[r["Counter_Name"]]
        k = order[int(r["Dispatch_Id"])]
        c[k] = c.get(k, 0.0) + float(r["Counter_Value"])
print(f"# bytes (or requests) per random 8-byte access, by randbench mode — {os.path.basename(d)}")
for ctr in sorted(vals):
    scale = 1024.0 if ctr in ("FETCH_SIZE", "WRITE_SIZE") else 1.0
    unit = "B/access" if scale != 1.0 else "req/access"
    for m in range(8):
        v = vals[ctr].get(2 * m + 1)
        if v is not None:
            print(f"{ctr:28s} mode {MODES[m]:24s} {v * scale / ACC[m]:8.2f} {unit}   (counter {v:.0f}, {ACC[m]} accesses)")
