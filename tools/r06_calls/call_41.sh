#!/bin/bash
# round 6, call 41: placement by measurement (seen_set_pick: kmc_open clears the table it got under HIP events and asks for another
# while holding it when the clear is below 6.8 TB/s, up to KMC_SEEN_SET_CANDIDATES = 4) against the first table as it comes (= 1):
# the tests of the seen-set's memory first; then the headline, fresh processes, interleaved; BASELINE config 4, config 5, the
# headline with traces; what it costs a CLI user (the front end's wall time).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r06_calls/call_41.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_41; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_seen_set_memory.py tests/test_gpu_sharded_and_traces.py -x -q -m gpu ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
export KMC_NO_TORCH=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-traces-leg --no-cold-start --no-baseline-configs --no-stretch --steps 3 --warmup 1"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; b = c.get('step_breakdown') or {}
        print('$1', 'k_expand %.2f ms, clear %.2f ms' % (b.get('k_expand_ms', 0), b.get('clear_seen_set_ms', 0)), 'golden', c['matches_oracle_golden'], 'open %.3f s' % (c.get('open_s') or -1), c.get('seen_set_placement'))
"; }
for rep in 1 2 3 4 5 6 7 8 9 10; do for k in 1 4; do
  KMC_SEEN_SET_CANDIDATES=$k KMC_VERBOSE=1 timeout 300 $B 2>$O/e.txt | pick "[rep $rep headline, candidates $k]" | tee -a $O/pick.txt; grep "placement" $O/e.txt | cut -c1-200 | tee -a $O/pick.txt
done; done
for rep in 1 2 3 4; do for k in 1 4; do
  KMC_SEEN_SET_CANDIDATES=$k timeout 300 $B --workload Kip279,5,2,2,1 2>/dev/null | pick "[rep $rep config 4, candidates $k]" | tee -a $O/pick.txt
  KMC_SEEN_SET_CANDIDATES=$k KMC_BENCH_TRACE=1 timeout 300 $B 2>/dev/null | pick "[rep $rep headline with traces, candidates $k]" | tee -a $O/pick.txt
done; done
T="kafka_specification_amd/tlc models/Kip320.tla -table 1073741824 -frontier 67108864 -v"
for rep in 1 2 3 4; do for k in 1 4; do for tr in "" "-notrace"; do
  s=$(date +%s.%N); KMC_SEEN_SET_CANDIDATES=$k $T $tr > $O/out.txt 2>/dev/null; e=$(date +%s.%N)
  echo "[front end, candidates $k ${tr:-traces}] wall $(python -c "print('%.3f' % ($e - $s))") s | $(grep -o 'allocation of [0-9.]* GiB [0-9.]*s' $O/out.txt) | $(grep -o 'search [0-9.]*s' $O/out.txt) | $(grep -o '[0-9.]*s teardown' $O/out.txt) | $(grep 'Seen-set placement' $O/out.txt)" | tee -a $O/pick.txt
done; done; done
