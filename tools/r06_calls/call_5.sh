#!/bin/bash
# round 6, call 5: the headline's k_expand came out bimodal in call 4 (28.9 or 31.3 ms at ONE table size, process to process).
# Is it where the allocation lands?  Fresh processes at three sizes, the table's address printed; the same with the table aligned to
# 1 GiB / 4 GiB inside a larger allocation; several handles inside one process.
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_5; mkdir -p $O
export KMC_NO_TORCH=1 KMC_VERBOSE=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-cold-start --no-baseline-configs --no-stretch"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; b = c.get('step_breakdown') or {}
        print('$1', 'ms/step %.2f' % j['ms_per_step'], 'k_expand %.2f clear %.2f' % (b.get('k_expand_ms', 0), b.get('clear_seen_set_ms', 0)), 'golden', c['matches_oracle_golden'])
"; }
G=$((1<<30))
for m in 1000 1500 2500; do for rep in 1 2 3 4 5; do
  KMC_BENCH_TABLE=$((G/1000*m)) timeout 300 $B --steps 5 --warmup 1 2>$O/e.txt | pick "[headline, $m/1000 x 2^30, process $rep]" | tee -a $O/placement.txt
  grep "seen-set" $O/e.txt | tail -1 | tee -a $O/placement.txt
done; done
for al in 30 32; do for m in 1000 1500; do for rep in 1 2 3; do
  KMC_TABLE_ALIGN_LOG2=$al KMC_BENCH_TABLE=$((G/1000*m)) timeout 300 $B --steps 5 --warmup 1 2>$O/e.txt | pick "[headline, $m/1000 x 2^30, aligned 2^$al, process $rep]" | tee -a $O/placement.txt
  grep "seen-set" $O/e.txt | tail -1 | tee -a $O/placement.txt
done; done; done
echo "== several handles in one process" | tee -a $O/placement.txt
python - <<'PY' 2>&1 | grep -v "^\[kmc\] spec" | tee -a $O/placement.txt
import os, sys
sys.path.insert(0, os.getcwd())
import kafka_specification_amd as kmc
from kafka_specification_amd.configs import HEADLINE
for size in (1 << 30, 3 << 29, 1 << 30, 3 << 29, 5 << 29):
    with kmc.ModelChecker(kmc.CheckerConfig(**HEADLINE, table_capacity=size, frontier_capacity=1 << 26)) as mc:
        mc.run()
        ks = [mc.run().seconds_expand * 1e3 for _ in range(4)]
        print(f"table {size / 2**30:.2f} x 2^30: k_expand {' '.join('%.2f' % k for k in ks)} ms", flush=True)
PY
