#!/bin/bash
# round 3, GPU call 18: the whole -m gpu suite with the grouped replica-major layout + kind-major walk as every Kafka
# configuration's default (tests/test_gpu_kind_major.py forces the other forms), then a bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c18; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|FAILED" $O/tests.log | tail -8
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-330 $O/bench.json
