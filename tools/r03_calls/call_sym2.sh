#!/bin/bash
# round 3, symmetry call 2: the suite on the table-driven permutations, then the headline under occupancy targets / dry passes
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests/test_gpu_symmetry.py -x -q > gpurun_out/s2/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s2/tests.log
tail -5 gpurun_out/s2/tests.log
export KMC_NO_TORCH=1
out=gpurun_out/s2/ablate.log
: > $out
run() { echo "== $1" >> $out; shift; env "$@" timeout 200 python tools/sym_headline.py 3 sym 2>&1 | grep -E "ms_step|dry/shadow" | tail -3 >> $out; }
run "default" A=1
run "KMC_DRYRUN=1" KMC_DRYRUN=1
for w in 6 5 4; do run "KMC_MIN_WAVES=$w" KMC_JIT_DEFINES=-DKMC_MIN_WAVES=$w; done
cat $out
