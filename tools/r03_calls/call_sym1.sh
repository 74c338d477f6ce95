#!/bin/bash
# round 3, symmetry call 1: the orbit-counting suite and the headline with / without symmetry
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/s1
timeout 900 python -m pytest tests/test_gpu_symmetry.py -x -q > gpurun_out/s1/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s1/tests.log
tail -25 gpurun_out/s1/tests.log
timeout 300 python tools/sym_headline.py 3 > gpurun_out/s1/headline.jsonl 2> gpurun_out/s1/headline.err
cat gpurun_out/s1/headline.jsonl; tail -3 gpurun_out/s1/headline.err
