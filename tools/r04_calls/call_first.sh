#!/bin/bash
# round 4, FIRST call (prepared at the end of round 3, never run): the one device change no GPU has executed — the 64-bit
# orbit-deficit cells of k_expand's block tail (kmc_device.h, kmc_corr) — then fresh orbit-counting profiles on it.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r04_calls/call_first.sh'
# Expect: tests_sym.log without an xfail (…deep_levels…[17] PASSED); config5_sym_L17.json states_generated 8992050881143.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04_first; mkdir -p $O
timeout 700 python -m pytest tests/test_gpu_symmetry.py -q -n 4 -rxX > $O/tests_sym.log 2>&1; echo "sym tests rc=$?" >> $O/tests_sym.log; tail -6 $O/tests_sym.log
( export KMC_NO_TORCH=1 KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
  for lv in 10 14 17; do
    timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget $lv --symmetry --no-cpu-baseline --steps 1 --warmup 0 > $O/config5_sym_L$lv.json 2> $O/config5_sym_L$lv.err
    python - $O/config5_sym_L$lv.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print({k:d[k] for k in ("value","ms_per_step")}, c["distinct_states"], c["states_generated"], c["depth"], c["verdict"], d["roofline"]["kernel_seconds_per_step"])
except Exception as e: print("no line:", e)
PY
  done
  cat $O/config5_sym_L10.json $O/config5_sym_L14.json $O/config5_sym_L17.json > $O/config5_orbit_counting.jsonl )   # -> profiles/r04_config5_orbit_counting.jsonl
PROFILE_BENCH_ARGS=--symmetry bash tools/profile.sh r04a_sym > $O/profile_sym.log 2>&1; tail -1 $O/profile_sym.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json     # traffic must be non-null (kernel_code_sha256)
timeout 1500 python -m pytest tests -m gpu -x -q -n 4 > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
