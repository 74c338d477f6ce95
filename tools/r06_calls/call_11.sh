#!/bin/bash
# round 6, call 11: the headline's two modes (k_expand 28.9 or 31.3 - 31.8 ms) once more, at the table sizes where the fast one has
# been seen (1.5 x and 2.5 x 2^30 slots).  (a) fresh processes, (b) SEVERAL HANDLES INSIDE ONE PROCESS, each opened while the earlier
# ones are alive - so each table lies somewhere else in the HBM - and then re-opened after closing all: does the mode follow the
# handle (placement) or the process?
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_11; mkdir -p $O
export KMC_NO_TORCH=1 KMC_VERBOSE=1
python - <<'PY' 2>&1 | grep -v "^\[kmc\] spec" | tee $O/handles.txt
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import kafka_specification_amd as kmc
from kafka_specification_amd.configs import HEADLINE
hip = ctypes.CDLL("libamdhip64.so")
def free_gb():
    f = ctypes.c_size_t(); t = ctypes.c_size_t(); hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)); return f.value / 2**30
for slots in (3 << 29, 5 << 29, 1 << 30):
    print(f"== table of {slots / 2**30:.2f} x 2^30 slots: handles opened one after the other, all kept alive", flush=True)
    alive = []
    for k in range(6):
        mc = kmc.ModelChecker(kmc.CheckerConfig(**HEADLINE, table_capacity=slots, frontier_capacity=1 << 26)).__enter__()
        mc.run()
        ks = [mc.run().seconds_expand * 1e3 for _ in range(3)]
        print(f"handle {k} (free {free_gb():.1f} GiB): k_expand {' '.join('%.2f' % x for x in ks)} ms", flush=True)
        alive.append(mc)
    print("-- the same handles again, in order (same placement, later in the process)", flush=True)
    for k, mc in enumerate(alive):
        ks = [mc.run().seconds_expand * 1e3 for _ in range(2)]
        print(f"handle {k}: k_expand {' '.join('%.2f' % x for x in ks)} ms", flush=True)
    for mc in alive:
        mc.__exit__(None, None, None)
PY
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-cold-start --no-baseline-configs --no-stretch"
for m in 3 5; do for rep in 1 2 3 4 5 6; do
  KMC_BENCH_TABLE=$((m<<29)) timeout 300 $B --steps 3 --warmup 1 2>$O/e.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); b = j['config'].get('step_breakdown') or {}
        print('[fresh process $rep, $m x 2^29 slots] k_expand %.2f clear %.2f golden %s' % (b.get('k_expand_ms', 0), b.get('clear_seen_set_ms', 0), j['config']['matches_oracle_golden']))
" | tee -a $O/processes.txt
done; done
