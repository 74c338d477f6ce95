#!/bin/bash
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/r02c27
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export KMC_LOOPBACK_EXCHANGES=rccl
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $REPO/tools/loopback_headline.py 8 > $OUT/run.log 2>&1
grep shards $OUT/run.log
ls $OUT/trace
