#!/bin/bash
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/r02c15
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/tools/loopback_headline.py 2 > $OUT/run.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-200
cat $OUT/run.log | tail -3
