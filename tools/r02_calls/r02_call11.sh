#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c11
mkdir -p $OUT
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; python3 -c "
import json; r=json.load(open('$OUT/bench.json')); print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline']['traffic'], json.dumps(r['roofline']['random_access']), r['cpu_baseline'])"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
