#!/bin/bash
# round 3, GPU call 10 (after the container was re-created: calls 8/9's outputs were lost): the whole -m gpu suite, the
# rocprofv3 passes of tools/profile.sh, a bench line, the ablation ladder, the 6.45 G-state stretch with 128-bit entries
# under three seeds, and the same configuration on 8 logical shards
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c10
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 900 tools/profile.sh r03 > $O/profile.log 2>&1; tail -3 $O/profile.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-260 $O/bench.json
rm -f gpurun_out/ablate.log; timeout 600 tools/ablate.sh > /dev/null 2>&1; cp gpurun_out/ablate.log $O/ablate.log; cat $O/ablate.log
KMC_NO_TORCH=1 timeout 600 python tools/fp128_stretch.py 0 0x5EED2 0xC0FFEE > $O/fp128_stretch.jsonl 2> $O/fp128_stretch.err
cat $O/fp128_stretch.jsonl; tail -2 $O/fp128_stretch.err
timeout 900 python tools/loopback_stretch.py 8 > $O/loopback_stretch.jsonl 2> $O/loopback_stretch.err
cat $O/loopback_stretch.jsonl; tail -2 $O/loopback_stretch.err
