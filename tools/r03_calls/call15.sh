#!/bin/bash
# round 3, GPU call 15: the whole -m gpu suite on the kind-major kernel (new: tests/test_gpu_kind_major.py forces each walk
# onto the other's configurations), then a bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c15; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=" $O/tests.log | tail -3
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-330 $O/bench.json
