#!/bin/bash
# round 3, GPU call 9: config 5 through the exchange (8 logical shards; 8 concurrent ranks over the mock transport), the
# whole suite, and Kip320 3/6/6/3 on 8 logical shards
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c9
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/c9/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/c9/tests.log
tail -6 gpurun_out/c9/tests.log
timeout 900 python tools/loopback_stretch.py 8 > gpurun_out/c9/loopback_stretch.jsonl 2> gpurun_out/c9/loopback_stretch.err
cat gpurun_out/c9/loopback_stretch.jsonl; tail -3 gpurun_out/c9/loopback_stretch.err
