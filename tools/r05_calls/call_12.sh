#!/bin/bash
# round 5, call 12: the -m gpu suite and smoke on the tree after the engine file was split into parts (text only)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_12; mkdir -p $O
ls kafka_specification_amd/kmc_cache | sort > $O/cache_before.txt
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
ls kafka_specification_amd/kmc_cache | sort > $O/cache_after.txt
echo "specialised on the box:"; comm -13 $O/cache_before.txt $O/cache_after.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
