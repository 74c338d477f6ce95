#!/bin/bash
# round 6, call 38: the seen-set's chunks created in 64 clusters with an unmapped spacer allocation after each (KMC_SEEN_SET_SPREAD = 4 / 16 / 32:
# the table lies over that many times its size) against the chunks as they come (=1): the headline, fresh processes, interleaved;
# what the pool costs at open (KMC_VERBOSE) and a CLI user (the front end's wall time); BASELINE config 4 and the headline with traces.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r06_calls/call_38.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_38; mkdir -p $O
export KMC_NO_TORCH=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-traces-leg --no-cold-start --no-baseline-configs --no-stretch --steps 3 --warmup 1"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; b = c.get('step_breakdown') or {}
        print('$1', 'k_expand %.2f ms, clear %.2f ms' % (b.get('k_expand_ms', 0), b.get('clear_seen_set_ms', 0)), 'golden', c['matches_oracle_golden'])
"; }
for rep in 1 2 3 4 5 6; do for f in 1 4 16 32; do
  KMC_SEEN_SET_SPREAD=$f KMC_VERBOSE=1 timeout 300 $B 2>$O/e.txt | pick "[rep $rep headline, spread x$f]" | tee -a $O/pool.txt; grep "chunks of" $O/e.txt | head -1 | cut -c1-200 | tee -a $O/pool.txt
done; done
for rep in 1 2 3; do for f in 1 4 16 32; do
  KMC_SEEN_SET_SPREAD=$f timeout 300 $B --workload Kip279,5,2,2,1 2>/dev/null | pick "[rep $rep config 4, spread x$f]" | tee -a $O/pool.txt
  KMC_SEEN_SET_SPREAD=$f KMC_BENCH_TRACE=1 timeout 300 $B 2>/dev/null | pick "[rep $rep headline with traces, spread x$f]" | tee -a $O/pool.txt
done; done
T="kafka_specification_amd/tlc models/Kip320.tla -table 1073741824 -frontier 67108864 -v"
for rep in 1 2 3 4; do for f in 1 4 16 32; do for tr in "" "-notrace"; do
  s=$(date +%s.%N); KMC_SEEN_SET_SPREAD=$f $T $tr > $O/out.txt 2>/dev/null; e=$(date +%s.%N)
  echo "[front end, spread x$f ${tr:-traces}] wall $(python -c "print('%.3f' % ($e - $s))") s | $(grep -o 'allocation of [0-9.]* GiB [0-9.]*s' $O/out.txt) | $(grep -o 'search [0-9.]*s' $O/out.txt) | $(grep -o '[0-9.]*s teardown' $O/out.txt)" | tee -a $O/pool.txt
done; done; done
