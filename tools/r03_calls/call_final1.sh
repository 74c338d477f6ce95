#!/bin/bash
# round 3, final call 1: tools/profile.sh on the plain headline (kernel stats + DRAM counters that carry this device code's
# sha), the same passes over the orbit-counting search (--symmetry), the default bench line quoting them, the orbit-counting
# kernel's ablation ladder, then the whole -m gpu suite on the tree with orbit counting for up to six replicas
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/f1; mkdir -p $O
bash tools/profile.sh r03f > $O/profile_plain.log 2>&1; tail -2 $O/profile_plain.log
PROFILE_BENCH_ARGS=--symmetry bash tools/profile.sh r03f_sym > $O/profile_sym.log 2>&1; tail -2 $O/profile_sym.log
# the bench line quotes the newest profiles/rNN_pmc_summary.json / rNN_summary.json when they carry this device code's sha
cp gpurun_out/prof_r03f/pmc_summary.json profiles/r03_pmc_summary.json; cp gpurun_out/prof_r03f/summary.json profiles/r03_summary.json
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
bash tools/sym_ablate.sh > /dev/null 2>&1; cp gpurun_out/sym_ablate.log $O/; tail -40 $O/sym_ablate.log
timeout 840 python -m pytest tests -v -m gpu -n 4 > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -E " passed| failed|rc=|FAILED|ERROR" $O/tests.log | tail -12
