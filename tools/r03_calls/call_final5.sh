#!/bin/bash
# round 3, final call 5: the round-end commands as the driver runs them — the -m gpu suite in ONE process (how long does it
# take without xdist?), smoke(), the default bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/f5; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=20 ) > $O/tests_serial.log 2>&1; echo "tests rc=$?" >> $O/tests_serial.log
grep -E " passed| failed|rc=|FAILED|ERROR|^real" $O/tests_serial.log | tail -8
grep -A22 "slowest" $O/tests_serial.log | head -24
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
( time python bench.py ) > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; tail -3 $O/bench.err
