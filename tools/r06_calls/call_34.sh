#!/bin/bash
# round 6, call 34: is the level of the headline's k_expand (28.4 / 29.5 - 30.3 / 31.7 ms, the clear moving with it) a property of the
# PROCESS or of the BOX AT THAT TIME?  One long-lived process (one handle) runs a search every ~6 s for five minutes; between its
# searches fresh processes run the same search (no torch / torch alternating).  The two never run at the same time (a lock file).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/r06_calls/call_34.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_34; mkdir -p $O; rm -f $O/go $O/done
KMC_NO_TORCH=1 python - > $O/long.txt 2>&1 <<'PY' &
import os, time
import kafka_specification_amd as kmc
O = "gpurun_out/r06_34"
c = dict(model="Kip320", n_replicas=3, log_size=6, max_records=6, max_leader_epoch=2, invariants=("TypeOk", "WeakIsr", "StrongIsr"))
t0 = time.time()
with kmc.ModelChecker(kmc.CheckerConfig(**c, device=0, table_capacity=1 << 30, frontier_capacity=1 << 26)) as mc:
    for k in range(48):
        while not os.path.exists(f"{O}/go"):      # my turn comes when the shell says so
            time.sleep(0.05)
        os.remove(f"{O}/go")
        xs = [mc.run() for _ in range(3)]
        print("long-lived handle, search %2d at %6.1f s: k_expand %.2f ms, clear %.2f ms" % (k, time.time() - t0, 1e3 * xs[-1].seconds_expand, 1e3 * xs[-1].seconds_clear), flush=True)
        open(f"{O}/done", "w").close()
PY
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-traces-leg --no-cold-start --no-baseline-configs --no-stretch --steps 3 --warmup 1"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); b = j['config'].get('step_breakdown') or {}
        print('$1', 'k_expand %.2f ms, clear %.2f ms' % (b.get('k_expand_ms', 0), b.get('clear_seen_set_ms', 0)))
"; }
s0=$(date +%s)
for k in $(seq 1 48); do
  touch $O/go; while [ ! -e $O/done ]; do sleep 0.05; done; rm -f $O/done
  tail -1 $O/long.txt | tee -a $O/series.txt
  if [ $((k % 2)) = 0 ]; then KMC_NO_TORCH=1 timeout 120 $B 2>/dev/null | pick "   fresh process (no torch) at $(( $(date +%s) - s0 )) s:" | tee -a $O/series.txt
  else timeout 120 $B 2>/dev/null | pick "   fresh process (torch)    at $(( $(date +%s) - s0 )) s:" | tee -a $O/series.txt; fi
done
wait
