#!/bin/bash
# round 3, GPU call 3: the whole -m gpu suite on the new device code (conservation counters, checksum, wide slots), what
# the always-on counters cost on the headline, the wide table's cost, bigger tables, and Kip320 3/6/6/3 with -fp128.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c3
rm -f gpurun_out/sweep.log
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/c3/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/c3/tests.log
tail -5 gpurun_out/c3/tests.log
tools/sweep.sh "base||" "base_again||"
KMC_BENCH_TABLE=$((1<<31)) tools/sweep.sh "table_16g||"
KMC_BENCH_TABLE=$((1<<32)) tools/sweep.sh "table_32g||"
KMC_BENCH_FP128=1 tools/sweep.sh "fp128_16g||"
KMC_BENCH_FP128=1 KMC_BENCH_TABLE=$((1<<31)) tools/sweep.sh "fp128_32g||"
cp gpurun_out/sweep.log gpurun_out/c3/sweep.log
KMC_NO_TORCH=1 timeout 600 python tools/fp128_stretch.py 0 0x5EED2 0xC0FFEE > gpurun_out/c3/fp128_stretch.jsonl 2> gpurun_out/c3/fp128_stretch.err
cat gpurun_out/c3/fp128_stretch.jsonl; tail -3 gpurun_out/c3/fp128_stretch.err
timeout 600 python bench.py > gpurun_out/c3/bench.json 2> gpurun_out/c3/bench.err
cut -c1-400 gpurun_out/c3/bench.json
