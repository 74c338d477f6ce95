#!/bin/bash
# round 3, symmetry call 3: the whole GPU suite on the tree with orbit counting, then the default bench line
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/s3
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/s3/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s3/tests.log
tail -6 gpurun_out/s3/tests.log
timeout 600 python bench.py > gpurun_out/s3/bench.json 2> gpurun_out/s3/bench.err
cat gpurun_out/s3/bench.json | cut -c1-6000; tail -3 gpurun_out/s3/bench.err
