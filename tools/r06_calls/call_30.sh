#!/bin/bash
# round 6, call 30: (1) the default bench line on a box that has run nothing before it (call 29's line, taken right after 21
# front-end processes had each released 11 - 19 GiB, read 33.6 ms per step where call 28's read 30.4 on the same code);
# (2) is it the memory other processes have just released?  The headline alone (fresh process, no torch): on the idle device; right
# after a process that held 200 GiB has ended; 20 s later; right after ten front-end processes back to back; 20 s later.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r06_calls/call_30.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_30; mkdir -p $O
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
j = json.load(open('gpurun_out/r06_30/bench.json'))
print('headline', round(j['ms_per_step'], 2), j['config'].get('step_breakdown'), j['config']['matches_oracle_golden'], round(j['roofline']['frac'], 4), j['roofline'].get('traffic') is not None)
print('traces_kept', j.get('traces_kept'))
print('orbit counting', (j.get('orbit_counting') or {}).get('ms_per_step'))
for k, v in j.get('baseline_configs', {}).items():
    print(k, round(v.get('ms_per_step', 0), 2), v.get('step_breakdown'), v.get('matches_oracle_golden'), (v.get('roofline') or {}).get('frac'), v.get('error'))
s = j.get('stretch_1gpu', {})
print('stretch', s.get('time_to_exhaustive_s'), s.get('matches_oracle_golden'), (s.get('roofline') or {}).get('frac'), s.get('error'))
print('cold_start', json.dumps(j.get('cold_start'))[:1200])
PY
export KMC_NO_TORCH=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-traces-leg --no-cold-start --no-baseline-configs --no-stretch --steps 5 --warmup 1"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; b = c.get('step_breakdown') or {}
        print('$1', 'ms/step %.2f' % j['ms_per_step'], 'k_expand %.2f clear %.2f' % (b.get('k_expand_ms', 0), b.get('clear_seen_set_ms', 0)), 'golden', c['matches_oracle_golden'])
"; }
hold() { python - <<'PY'
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
ps = []
for k in range(25):                       # 25 x 8 GiB, touched: a process that used 200 GiB of the device
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(8 << 30)) == 0
    hip.hipMemset(p, 0, ctypes.c_size_t(8 << 30))
    ps.append(p)
hip.hipDeviceSynchronize()
PY
}
sleep 20
for rep in 1 2; do
  timeout 300 $B 2>>$O/err.txt | pick "[rep $rep: idle device]" | tee -a $O/release.txt
  hold
  timeout 300 $B 2>>$O/err.txt | pick "[rep $rep: right after a process that held 200 GiB ended]" | tee -a $O/release.txt
  timeout 300 $B 2>>$O/err.txt | pick "[rep $rep: ... the next process]" | tee -a $O/release.txt
  sleep 20
  timeout 300 $B 2>>$O/err.txt | pick "[rep $rep: 20 s later]" | tee -a $O/release.txt
  for k in 1 2 3 4 5 6 7 8 9 10; do kafka_specification_amd/tlc models/Kip320.tla -table 1073741824 -frontier 67108864 > /dev/null 2>&1; done
  timeout 300 $B 2>>$O/err.txt | pick "[rep $rep: right after ten front-end processes (19 GiB each) back to back]" | tee -a $O/release.txt
  sleep 20
  timeout 300 $B 2>>$O/err.txt | pick "[rep $rep: 20 s later]" | tee -a $O/release.txt
done
tail -3 $O/err.txt
