#!/bin/bash
# round 6, call 29: what a CLI user waits for, after the chunked seen-set: the native front end leaves without unmapping its chunks
# (default) against the full teardown (KMC_CLI_TEARDOWN=1), with and without traces, fresh processes, interleaved, five times; where
# a full teardown's time goes (KMC_VERBOSE: unmap / address range); then the default bench line once more (the traces_kept leg, the
# cold_start leg of the new front end) -> profiles/r06_bench.json.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r06_calls/call_29.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_29; mkdir -p $O
T="kafka_specification_amd/tlc models/Kip320.tla -table 1073741824 -frontier 67108864 -v"
$T > /dev/null 2>&1   # (the box's first process pages the libraries in)
for rep in 1 2 3 4 5; do for td in 0 1; do for tr in "" "-notrace"; do
  s=$(date +%s.%N)
  KMC_CLI_TEARDOWN=$td KMC_VERBOSE=1 $T $tr > $O/out.txt 2> $O/errv.txt; rc=$?
  e=$(date +%s.%N)
  echo "[teardown=$td ${tr:-traces}] rc $rc wall $(python -c "print('%.3f' % ($e - $s))") s | $(grep -o 'Wall time: [0-9.]*s' $O/out.txt) $(grep -o '[0-9.]*s teardown' $O/out.txt) | $(grep -c 'distinct states found' $O/out.txt) result line(s) | $(grep 'released' $O/errv.txt | tr '\n' ';')" | tee -a $O/cold.txt
done; done; done
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
j = json.load(open('gpurun_out/r06_29/bench.json'))
print('headline', round(j['ms_per_step'], 2), j['config']['matches_oracle_golden'], round(j['roofline']['frac'], 4), j['roofline'].get('traffic') is not None)
print('traces_kept', j.get('traces_kept'))
for k, v in j.get('baseline_configs', {}).items():
    print(k, round(v.get('ms_per_step', 0), 2), v.get('matches_oracle_golden'), (v.get('roofline') or {}).get('traffic') is not None, v.get('error'))
s = j.get('stretch_1gpu', {})
print('stretch', s.get('time_to_exhaustive_s'), s.get('matches_oracle_golden'), (s.get('roofline') or {}).get('traffic') is not None, s.get('error'))
print('cold_start', json.dumps(j.get('cold_start'))[:1200])
PY
tail -3 $O/bench.err
